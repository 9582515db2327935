#!/bin/bash
# One counter pass over the bench workload: tools/pmc_one.sh OUTDIR NAME COUNTER...  (at most 4 TCP_ counters a pass)
set -e
OUT=$(realpath -m "$1"); NAME=$2; shift 2
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export RT_OVERLAP=0
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/$NAME" --output-format csv -- python3 "$REPO/bench.py" --no-cpu-baseline --sustain-seconds 0 --steps 6 --warmup 1 > "$OUT/$NAME.log" 2>&1
python3 "$REPO/tools/pmc_summary.py" "$OUT/$NAME" "${RT_PMC_KERNEL:-trace_kernel<1, false, false, false>}"
