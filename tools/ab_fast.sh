#!/bin/bash
# A/B of library variants in fast mode: tools/ab_fast.sh name1 name2 ...   ("default" or build/ab/<name>.so)
mkdir -p gpurun_out
for n in "$@"; do
  if [ "$n" = default ]; then L=raytracer_amd/librt_mi355x.so; else L=build/ab/$n.so; fi
  RT_MI355X_LIB=$L timeout -k 10 120 python3 bench.py --no-cpu-baseline --sustain-seconds 0 --steps 30 --warmup 5 > gpurun_out/abf_$n.json 2> gpurun_out/abf_$n.err || { echo "$n FAILED"; tail -3 gpurun_out/abf_$n.err; exit 1; }
  python3 - "$n" <<'PY'
import json, sys
n = sys.argv[1]
d = json.loads(open(f"gpurun_out/abf_{n}.json").read().strip().splitlines()[-1])
f = d["fast_mode"]
print(n, "fast step", round(f["ms_per_step"], 4), "ms  fast serial launch", round(f["kernel_ms"], 4), "ms  nodes/ray", round(f["node_visits_per_ray"], 2),
      " | parity step", round(d["ms_per_step"], 4), "ms  ratio", round(d["ms_per_step"] / f["ms_per_step"], 3))
PY
done
