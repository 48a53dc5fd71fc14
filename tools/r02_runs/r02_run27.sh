#!/bin/bash
# thin-slab share (8192 x 1024, 80 iterations) after the alternating stream direction: fuse depth x rows per tile
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run27
mkdir -p $O
B="python bench.py --no-cpu-baseline --sim-steps 0 --steps 30 --warmup 5 --dim-y 1024"
for fuse in 8 10 12 14 16; do for rows in 0 32 44 64 96; do
  $B --fuse $fuse --sor-rows $rows > $O/f${fuse}_r$rows.json 2>> $O/err.log
  python - <<PY
import json
d = json.load(open("$O/f${fuse}_r$rows.json"))
print("fuse %2d rows %3d: %.4f ms  launches %2d  %.2f us/launch" % ($fuse, $rows, d["ms_per_step"], d["config"]["sor_launches_per_solve"], d["roofline"]["avg_launch_us"]))
PY
done; done
