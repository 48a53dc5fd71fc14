#!/bin/bash
# diagnostic builds of the NS = 16 / dx = 1 kernels: without global memory traffic, without LDS traffic (results wrong by design)
set -u
export TMPDIR=/tmp
for v in base diag3 diag4 diag5 base diag3 diag4 diag5; do
  L=""; [ $v != base ] && L=$PWD/esp32-fluid-simulation_amd/lib/variants/libsfl_hip_$v.so
  SFL_LIB=$L python bench.py --no-cpu-baseline --sim-steps 0 --steps 30 --warmup 5 > gpurun_out/r02_run44.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r02_run44.json"))
print("%-6s %.4f ms  %.2f us/launch" % ("$v", d["ms_per_step"], d["roofline"]["avg_launch_us"]))
PY
done
