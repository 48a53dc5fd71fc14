#!/bin/bash
# boundary-tile balance factor after the rhs read-ahead: sweep
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run40
mkdir -p $O
for cfg in "full:" "s2048:--dim-y 2048" "s1024:--dim-y 1024" "c5:--size 16384 --iters 200 --steps 5"; do
  n=${cfg%%:*}; a=${cfg#*:}
  for k in 8 9 10 11 12 13 10; do
    SFL_EDGE_COST16=$k python bench.py --no-cpu-baseline --sim-steps 0 --steps 30 --warmup 5 $a > $O/b.json 2>> $O/err.log
    python - <<PY
import json
d = json.load(open("$O/b.json"))
print("%-6s edge cost %2d/16: %.4f ms  %.2f us/launch" % ("$n", $k, d["ms_per_step"], d["roofline"]["avg_launch_us"]))
PY
  done
done
