#!/bin/bash
# Everything the committed profiles of a round come from (run on the GPU box):
#   tools/profile_round.sh OUTDIR
# 1. rocprofv3 --kernel-trace --stats of the default bench command, one frame at a time (RT_OVERLAP=0)
#    and with the default two frames in flight;  2. the counter passes (tools/pmc_passes.sh).
set -e
OUT=$(realpath -m "$1")
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
RT_OVERLAP=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/serial" --output-format csv -- python3 "$REPO/bench.py" --no-cpu-baseline --sustain-seconds 0 --steps 20 --warmup 3 > "$OUT/serial_bench.json" 2> "$OUT/serial.err"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/overlap" --output-format csv -- python3 "$REPO/bench.py" --no-cpu-baseline --sustain-seconds 0 --steps 20 --warmup 3 > "$OUT/overlap_bench.json" 2> "$OUT/overlap.err"
cd "$REPO"
for k in serial overlap; do
  f=$(ls "$OUT"/$k/*/*kernel_stats.csv | head -1); cp "$f" "$OUT/${k}_kernel_stats.csv"
done
tools/pmc_passes.sh "$OUT/pmc" > "$OUT/pmc.log" 2>&1
cat "$OUT"/pmc/*.txt > "$OUT/pmc_summary.txt"
head -5 "$OUT/serial_kernel_stats.csv"
