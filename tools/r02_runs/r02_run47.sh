#!/bin/bash
set -u
export TMPDIR=/tmp
for lc in 2 4 2 4; do
  SFL_LIB=$PWD/esp32-fluid-simulation_amd/lib/variants/libsfl_hip_l4noedge.so python bench.py --no-cpu-baseline --sim-steps 0 --steps 30 --warmup 5 --lane-cells $lc > gpurun_out/r02_run47.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r02_run47.json"))
print("lane cells $lc (no EDGE path in the 4-cell kernel): %.4f ms  %.2f us/launch" % (d["ms_per_step"], d["roofline"]["avg_launch_us"]))
PY
done
