#!/bin/bash
# Counter passes over the bench workload, one rocprofv3 run per pass (run on the GPU box), each with
# --kernel-trace so that the kernel's duration in that very run is known.
#   tools/pmc_passes.sh OUTDIR [extra bench args]
# Each pass respects the per-block counter limits (8 SQ, 4 TCC with FETCH_SIZE = 3 and WRITE_SIZE = 2,
# 4 TCP, 2 TA, 2 TD, 2 GRBM).  tools/pmc_to_json.py turns OUTDIR into profiles/pmc_trace_kernel.json.
set -e
OUT=$(realpath -m "$1"); shift
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export RT_OVERLAP=0
run() {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/$name" --output-format csv -- python3 "$REPO/bench.py" --no-cpu-baseline --sustain-seconds 0 --steps 6 --warmup 1 "${EXTRA[@]}" > "$OUT/$name.log" 2>&1
  python3 "$REPO/tools/pmc_summary.py" "$OUT/$name" "trace_kernel<1, false, false, false>" > "$OUT/$name.txt"
  cat "$OUT/$name.txt"
}
EXTRA=("$@")
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_FLAT GRBM_GUI_ACTIVE
run sq3 SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE
run tcc1 FETCH_SIZE TCC_HIT_sum GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE
