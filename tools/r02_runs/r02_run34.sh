#!/bin/bash
# 2- and 4-GPU shares of 8192^2 (4096 / 2048 rows) and the 8-GPU share of 16384^2: fuse depth sweep
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run34
mkdir -p $O
for cfg in "s2048:--dim-y 2048" "s4096:--dim-y 4096" "c5s:--size 16384 --dim-y 2048 --iters 200 --steps 8" "s1536:--dim-y 1536" "s3072:--dim-y 3072"; do
  n=${cfg%%:*}; a=${cfg#*:}
  for fuse in 10 12 14 16; do
    python bench.py --no-cpu-baseline --sim-steps 0 --steps 20 --warmup 5 $a --fuse $fuse > $O/${n}_f$fuse.json 2>> $O/err.log
    python - <<PY
import json
d = json.load(open("$O/${n}_f$fuse.json"))
print("%-6s fuse %2d: %.4f ms  launches %2d  %.2f us/launch" % ("$n", $fuse, d["ms_per_step"], d["config"]["sor_launches_per_solve"], d["roofline"]["avg_launch_us"]))
PY
  done
done
