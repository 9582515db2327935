#!/bin/bash
# Which nodes form the workgroups' LDS copy: the breadth-first top of the tree (RT_NODE_LAYOUT=0) against the
# largest boxes grown from the root (3, the default).  Bench timing of each, twice, in one run (figures move by
# about 1 % from box to box, so only runs of one call compare), then the parity tests of the default pipeline.
#   tools/ab_top.sh        (run on the GPU box)
set -e
OUT=gpurun_out/l3; mkdir -p $OUT
for L in 0 3 0 3; do
  export RT_NODE_LAYOUT=$L
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --sustain-seconds 0 --steps 30 --warmup 5 > $OUT/layout$L.json 2> $OUT/layout$L.err
  python3 - $OUT/layout$L.json $L <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("layout", sys.argv[2], "step", round(d["ms_per_step"], 4), "ms  reference frame", round(d["reference_frame"]["ms_per_step"], 4),
      "ms  serial launch", round(d["roofline"]["kernel_ms"], 4), "ms", "fast", round(d["fast_mode"]["ms_per_step"],4), flush=True)
PY
done
RT_NODE_LAYOUT=3 timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py -x -q -k "mega and (colour or primary or bench_workload)" --timeout 200 --timeout-method thread > $OUT/parity3.log 2>&1 || { echo "parity FAILED"; tail -5 $OUT/parity3.log; exit 1; }
tail -1 $OUT/parity3.log
