#!/usr/bin/env python3
"""Collects the per-pass summaries written by tools/pmc_passes.sh into the JSON bench.py reads
(profiles/pmc_trace_kernel.json):  tools/pmc_to_json.py PASSDIR WORKLOAD_KEY OUT.json [note]"""
import json
import re
import sys
from pathlib import Path

d, key, out = Path(sys.argv[1]), sys.argv[2], Path(sys.argv[3])
note = sys.argv[4] if len(sys.argv) > 4 else ""
counters, durations, per_pass = {}, [], {}
for f in sorted(d.glob("*.txt")):
    for line in f.read_text().splitlines():
        m = re.match(r"(\S+)\s+n=\s*(\d+)\s+mean=(\S+)", line)
        if not m:
            continue
        name, val = m.group(1), float(m.group(3))
        if name == "KERNEL_DURATION_NS":
            durations.append(val)
            per_pass[f.stem] = val / 1e6
        elif name == "GRBM_GUI_ACTIVE":
            counters.setdefault("GRBM_GUI_ACTIVE_per_pass", {})[f.stem] = val
        else:
            counters[name] = val
g = counters.pop("GRBM_GUI_ACTIVE_per_pass", {})
counters["GRBM_GUI_ACTIVE"] = sum(g.values()) / max(1, len(g))
rec = {"kernel": "rt::trace_kernel<1, false, false, false>", "workload_key": key,
       "kernel_ms_in_pmc_run": sum(durations) / len(durations) / 1e6, "kernel_ms_per_pass": per_pass,
       "valu_issue_cycles": 2.45,
       "valu_issue_cycles_source": "tools/microbench/valu_rate.hip on MI355X: a SIMD issues one independent wave64 "
                                   "fp32 instruction per 2.4-2.55 cycles from two or more waves (5.0 from one wave alone)",
       "counters": counters, "units": "FETCH_SIZE / WRITE_SIZE in KB; SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* in quad-cycles; "
                                      "GRBM_GUI_ACTIVE summed over the 8 XCDs; per launch (mean over the launches of a pass)",
       "source": "RT_OVERLAP=0 tools/pmc_passes.sh (separate rocprofv3 --kernel-trace --pmc passes over bench.py "
                 "--no-cpu-baseline --steps 6 --warmup 1), MI355X", "note": note}
out.write_text(json.dumps(rec, indent=1) + "\n")
print(out, "written;", len(counters), "counters; kernel", round(rec["kernel_ms_in_pmc_run"], 4), "ms")
