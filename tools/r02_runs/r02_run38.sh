#!/bin/bash
# shares of the 8192^2 and 16384^2 solves with the final round-2 library (scaling table of DESIGN section 6)
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run38
mkdir -p $O
B="python bench.py --no-cpu-baseline --sim-steps 0 --steps 30 --warmup 5"
for rep in 1 2; do
for cfg in "full:" "s4096:--dim-y 4096" "s2048:--dim-y 2048" "s1024:--dim-y 1024" "c2:--size 2048 --iters 40" "c5:--size 16384 --iters 200 --steps 6" "c5s:--size 16384 --dim-y 2048 --iters 200 --steps 8" "c1:--size 61 --dim-y 81 --iters 20 --steps 200"; do
  n=${cfg%%:*}; a=${cfg#*:}
  $B $a > $O/${n}_$rep.json 2>> $O/err.log
  python - <<PY
import json
d = json.load(open("$O/${n}_$rep.json"))
print("%-6s %.4f ms  %.3e cell-iters/s  fuse %2d launches %2d  %.2f us/launch" % ("$n", d["ms_per_step"], d["value"], d["config"]["half_sweeps_fused_per_launch"], d["config"]["sor_launches_per_solve"], d["roofline"]["avg_launch_us"]))
PY
done; done
