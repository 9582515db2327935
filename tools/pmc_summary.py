#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per-kernel mean of each counter (one row per dispatch and
counter), and the kernel's mean duration when the run also has a kernel trace."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "trace_kernel<1, false, false, false>"
acc = collections.defaultdict(list)
for fn in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(fn) as f:
        for row in csv.DictReader(f):
            if kern in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for fn in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    with open(fn) as f:
        for row in csv.DictReader(f):
            if kern in row["Kernel_Name"]:
                acc["KERNEL_DURATION_NS"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:40s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
