#!/bin/bash
# Three counter passes (instruction mix, wave states, vector-memory path) over the bench workload,
# summarised for both the default and the fast-mode trace kernel:  tools/pmc_quick.sh OUTDIR
set -e
OUT=$(realpath -m "$1"); shift
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export RT_OVERLAP=0
# kernels to summarise: RT_PMC_KERNELS="name1;name2" overrides the default pair
if [ -n "$RT_PMC_KERNELS" ]; then IFS=';' read -r -a KERNELS <<< "$RT_PMC_KERNELS"
else KERNELS=("trace_regroup_kernel<1, false, false>" "trace_kernel<1, false, false, false>" "trace_kernel<1, false, false, true>"); fi
run() {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/$name" --output-format csv -- python3 "$REPO/bench.py" --no-cpu-baseline --sustain-seconds 0 --steps 6 --warmup 1 > "$OUT/$name.log" 2>&1
  for k in "${KERNELS[@]}"; do
    echo "== $name $k"; python3 "$REPO/tools/pmc_summary.py" "$OUT/$name" "$k"
  done
}
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
