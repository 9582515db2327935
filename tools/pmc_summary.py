#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per-kernel mean of each counter (one row per dispatch and counter)."""
import csv, sys, glob, collections
root = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "trace_kernel<1"
acc = collections.defaultdict(list)
for fn in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(fn) as f:
        for row in csv.DictReader(f):
            if kern in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:40s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
