#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run14
mkdir -p $O
B="python bench.py --no-cpu-baseline --sim-steps 0 --steps 20 --warmup 5"
run() { n=$1; shift; $B "$@" > $O/$n.json 2>> $O/bench.err; python - <<PY
import json
try:
    d=json.load(open("$O/$n.json"))
    print("%-30s %8.4f ms  fuse %2d launches %2d  %7.2f us/launch" % ("$n", d["ms_per_step"], d["config"]["half_sweeps_fused_per_launch"], d["config"]["sor_launches_per_solve"], d["roofline"]["avg_launch_us"]))
except Exception as e:
    print("$n", "FAILED", e)
PY
}
for fuse in 8 10; do for rows in 0 36 40 48 52 56; do run slab1024_f${fuse}_r$rows --dim-y 1024 --fuse $fuse --sor-rows $rows; done; done
for rows in 52 56 60 68 72 80; do run slab1024_f16_r$rows --dim-y 1024 --fuse 16 --sor-rows $rows; done
for fuse in 10 12 14 16; do run slab2048_f${fuse}_r0 --dim-y 2048 --fuse $fuse; done
for rows in 64 86 100 128; do run slab2048_f16_r$rows --dim-y 2048 --fuse 16 --sor-rows $rows; run slab2048_f12_r$rows --dim-y 2048 --fuse 12 --sor-rows $rows; done
for fuse in 12 14 16; do run slab4096_f${fuse}_r0 --dim-y 4096 --fuse $fuse; done
