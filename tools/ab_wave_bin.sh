#!/bin/bash
# Wavefront pipeline with and without regrouping the rays of a tile by direction octant between
# bounces (RT_WAVE_BIN): parity, timing, and the traversal kernel's instruction counters.
#   tools/ab_wave_bin.sh OUTDIR        (run on the GPU box)
set -e
OUT=$(realpath -m "$1")
mkdir -p "$OUT"
export RT_PIPELINE=wave
RT_WAVE_BIN=1 timeout -k 10 400 python3 -u -m pytest tests/test_gpu_parity.py -x -q -k "wave" --timeout 200 --timeout-method thread > "$OUT/parity_bin.log" 2>&1 || { echo "parity FAILED with RT_WAVE_BIN=1"; tail -8 "$OUT/parity_bin.log"; exit 1; }
tail -1 "$OUT/parity_bin.log"
for B in 0 1; do
  export RT_WAVE_BIN=$B
  RT_OVERLAP=0 timeout -k 10 200 python3 bench.py --no-cpu-baseline --sustain-seconds 0 --steps 30 --warmup 5 > "$OUT/bin$B.json" 2> "$OUT/bin$B.err"
  python3 - "$OUT/bin$B.json" $B <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("RT_WAVE_BIN", sys.argv[2], "step", round(d["ms_per_step"], 4), "ms  reference frame", round(d["reference_frame"]["ms_per_step"], 4), "ms", flush=True)
PY
  RT_PMC_KERNEL="extend_kernel<false>" tools/pmc_one.sh "$OUT" sq$B SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE
  RT_PMC_KERNEL="extend_kernel<false>" tools/pmc_one.sh "$OUT" ta$B TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
  RT_PMC_KERNEL="shade_kernel<1, false, false" tools/pmc_one.sh "$OUT" sh$B GRBM_GUI_ACTIVE
done
