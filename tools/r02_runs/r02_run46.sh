#!/bin/bash
set -u
export TMPDIR=/tmp
python tests/l4_probe.py 2>&1 | tail -9
for lc in 2 4 2 4; do
  python bench.py --no-cpu-baseline --sim-steps 0 --steps 30 --warmup 5 --lane-cells $lc > gpurun_out/r02_run46.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r02_run46.json"))
print("lane cells $lc: %.4f ms  %.2f us/launch" % (d["ms_per_step"], d["roofline"]["avg_launch_us"]))
PY
done
