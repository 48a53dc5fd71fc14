#!/bin/bash
# taller tiles (fewer waves per SIMD, fewer warm-up rows) on the whole domain
set -u
export TMPDIR=/tmp
for rows in 0 200 234 280 320 351 400 470 0; do
  python bench.py --no-cpu-baseline --sim-steps 0 --steps 30 --warmup 5 --sor-rows $rows > gpurun_out/r02_run45.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r02_run45.json"))
print("rows %3d: %.4f ms  %.2f us/launch" % ($rows, d["ms_per_step"], d["roofline"]["avg_launch_us"]))
PY
done
