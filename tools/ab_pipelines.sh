#!/bin/bash
# Times the bench workload with the megakernel and with the wavefront pipeline (same library):
#   tools/ab_pipelines.sh TAG     -> gpurun_out/TAG_{mega,wave}.json
set -e
TAG=${1:-ab}
mkdir -p gpurun_out
for p in mega wave; do
  RT_PIPELINE=$p timeout -k 10 120 python3 bench.py --no-cpu-baseline --sustain-seconds 0 --steps 30 --warmup 5 > gpurun_out/${TAG}_$p.json 2> gpurun_out/${TAG}_$p.err
  python3 - "$p" "gpurun_out/${TAG}_$p.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "step", round(d["ms_per_step"], 4), "ms  reference frame", round((d.get("reference_frame") or {}).get("ms_per_step", 0), 4),
      "ms  serial launch", round(d["roofline"]["kernel_ms"], 4), "ms")
PY
done
