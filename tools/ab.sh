#!/bin/bash
# A/B timing of library variants on the bench workload: tools/ab.sh name1 name2 ...
# ("default" = raytracer_amd/librt_mi355x.so, anything else = build/ab/<name>.so)
mkdir -p gpurun_out
for n in "$@"; do
  if [ "$n" = default ]; then L=raytracer_amd/librt_mi355x.so; else L=build/ab/$n.so; fi
  RT_MI355X_LIB=$L timeout -k 10 120 python3 bench.py --no-cpu-baseline --sustain-seconds 0 --steps 30 --warmup 5 > gpurun_out/ab_$n.json 2> gpurun_out/ab_$n.err || { echo "$n FAILED"; tail -3 gpurun_out/ab_$n.err; exit 1; }
  python3 - "$n" <<'PY'
import json, sys
n = sys.argv[1]
d = json.loads(open(f"gpurun_out/ab_{n}.json").read().strip().splitlines()[-1])
print(n, "step", round(d["ms_per_step"], 4), "ms  reference frame", round((d.get("reference_frame") or {}).get("ms_per_step", 0), 4), "ms  serial launch",
      round(d["roofline"]["kernel_ms"], 4), "ms")
PY
done
