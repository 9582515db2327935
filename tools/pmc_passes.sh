#!/bin/bash
# Counter passes over the bench workload, one rocprofv3 run per pass (run on the GPU box).
#   tools/pmc_passes.sh OUTDIR [extra bench args]
# Each pass respects the per-block counter limits; summaries are printed by tools/pmc_summary.py.
set -e
OUT=$(realpath -m "$1"); shift
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" -d "$OUT/$name" --output-format csv -- python3 "$REPO/bench.py" --no-cpu-baseline --steps 6 --warmup 1 $EXTRA > "$OUT/$name.log" 2>&1
  python3 "$REPO/tools/pmc_summary.py" "$OUT/$name" "trace_kernel<1" > "$OUT/$name.txt"
  cat "$OUT/$name.txt"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAIT_INST_LDS
run sq3 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_IFETCH SQ_INSTS_SMEM
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE GRBM_COUNT
run tcc1 FETCH_SIZE TCC_HIT_sum
run tcc2 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
