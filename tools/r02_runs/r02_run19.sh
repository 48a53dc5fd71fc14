#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run19
mkdir -p $O
B="python bench.py --no-cpu-baseline --sim-steps 0 --steps 40 --warmup 10"
run() { n=$1; shift; $B "$@" > $O/$n.json 2>> $O/bench.err; python - <<PY
import json
try:
    d=json.load(open("$O/$n.json"))
    print("%-30s %8.4f ms  fuse %2d launches %2d  %7.2f us/launch" % ("$n", d["ms_per_step"], d["config"]["half_sweeps_fused_per_launch"], d["config"]["sor_launches_per_solve"], d["roofline"]["avg_launch_us"]))
except Exception as e:
    print("$n", "FAILED", e)
PY
}
for fuse in 6 8 10 12 16; do run c2_2048_f$fuse --size 2048 --iters 40 --fuse $fuse; done
for fuse in 8 10 12; do run s1024sq_f$fuse --size 1024 --iters 40 --fuse $fuse; done
for fuse in 8 10 12 14 16; do run s3072sq_f$fuse --size 3072 --iters 40 --fuse $fuse; done
for fuse in 10 12 14 16; do run s4096sq_f$fuse --size 4096 --iters 40 --fuse $fuse; done
for fuse in 8 10 12; do run slab512_f$fuse --dim-y 512 --fuse $fuse; run slab768_f$fuse --dim-y 768 --fuse $fuse; done
for fuse in 10 12 14 16; do run slab1536_f$fuse --dim-y 1536 --fuse $fuse; done
