#!/bin/bash
# Wavefront pipeline over refill thresholds (lanes that must be idle before a wavefront refills from the queue)
for R in 8 16 32 48; do RT_PIPELINE=wave RT_REFILL=$R timeout -k 10 120 python3 bench.py --no-cpu-baseline --sustain-seconds 0 --steps 20 --warmup 3 > gpurun_out/sw.json 2>gpurun_out/sw.err; python3 -c "
import json
d=json.loads(open('gpurun_out/sw.json').read().strip().splitlines()[-1])
print('refill $R serial', round(d['roofline']['kernel_ms'],3), 'step', round(d['ms_per_step'],3))"; done
