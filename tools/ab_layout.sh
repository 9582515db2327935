#!/bin/bash
# Node-record layouts below the LDS copy (RT_NODE_LAYOUT = 0 breadth-first, 1 depth-first, 2 treelets):
# bench timing, parity tests of the default pipeline, and the cache / data-return counters of each.
#   tools/ab_layout.sh OUTDIR        (run on the GPU box)
set -e
OUT=$(realpath -m "$1")
mkdir -p "$OUT"
for L in 0 1 2; do
  export RT_NODE_LAYOUT=$L
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --sustain-seconds 0 --steps 30 --warmup 5 > "$OUT/layout$L.json" 2> "$OUT/layout$L.err"
  python3 - "$OUT/layout$L.json" $L <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("layout", sys.argv[2], "step", round(d["ms_per_step"], 4), "ms  reference frame", round(d["reference_frame"]["ms_per_step"], 4),
      "ms  serial launch", round(d["roofline"]["kernel_ms"], 4), "ms", flush=True)
PY
  timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py -x -q -k "mega and (colour or primary or bench_workload)" --timeout 200 --timeout-method thread > "$OUT/parity$L.log" 2>&1 || { echo "parity FAILED for layout $L"; tail -5 "$OUT/parity$L.log"; exit 1; }
  tail -1 "$OUT/parity$L.log"
  tools/pmc_one.sh "$OUT" tcc$L TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE > "$OUT/tcc$L.txt"
  tools/pmc_one.sh "$OUT" ta$L TA_TA_BUSY_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE > "$OUT/ta$L.txt"
  cat "$OUT/tcc$L.txt" "$OUT/ta$L.txt"
done
