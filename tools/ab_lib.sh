#!/bin/bash
# A/B of library builds on the parity path: tools/ab_lib.sh name1 name2 ...  ("default" or build/ab/<name>.so)
# For every build other than "default" the parity, model and no-allocation suites run first; then the
# steady frame times of C2, C3 and the bench scene, and the bench step.   (run on the GPU box)
mkdir -p gpurun_out
for n in "$@"; do
  if [ "$n" = default ]; then L=$PWD/raytracer_amd/librt_mi355x.so; else L=$PWD/build/ab/$n.so; fi
  export RT_MI355X_LIB=$L
  if [ "$n" != default ]; then
    timeout -k 10 900 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py tests/test_zz_no_alloc.py -x -q --timeout 300 --timeout-method thread > gpurun_out/ab_$n.parity.log 2>&1 || { echo "$n: parity FAILED"; tail -15 gpurun_out/ab_$n.parity.log; exit 1; }
    echo "$n: $(tail -1 gpurun_out/ab_$n.parity.log)"
  fi
  timeout -k 10 200 python3 tools/ab_small.py > gpurun_out/ab_$n.small.json 2> gpurun_out/ab_$n.err || { echo "$n: ab_small FAILED"; tail -5 gpurun_out/ab_$n.err; exit 1; }
  echo "$n: $(tail -1 gpurun_out/ab_$n.small.json)"
  RT_OVERLAP=0 timeout -k 10 200 python3 bench.py --no-cpu-baseline --sustain-seconds 0 --steps 40 --warmup 5 > gpurun_out/ab_$n.bench.json 2>> gpurun_out/ab_$n.err || { echo "$n: bench FAILED"; tail -5 gpurun_out/ab_$n.err; exit 1; }
  python3 - "$n" <<'PY'
import json, sys
n = sys.argv[1]
d = json.loads(open(f"gpurun_out/ab_{n}.bench.json").read().strip().splitlines()[-1])
print(n, "serial: step", round(d["ms_per_step"], 4), "ms  launch", round(d["roofline"].get("kernel_ms", 0), 4), "ms", flush=True)
PY
done
