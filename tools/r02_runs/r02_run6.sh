#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run6
mkdir -p $O
( python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "twin or randomised or config2" ) > $O/pytest_twin.log 2>&1
tail -3 $O/pytest_twin.log
B="python bench.py --no-cpu-baseline --sim-steps 0 --steps 20 --warmup 5"
run() { # name args...
  n=$1; shift
  $B "$@" > $O/$n.json 2>> $O/bench.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$n.json"))
    print("%-34s %8.4f ms  fuse %2d launches %2d  %7.2f us/launch  %.3e" % ("$n", d["ms_per_step"], d["config"]["half_sweeps_fused_per_launch"], d["config"]["sor_launches_per_solve"], d["roofline"]["avg_launch_us"], d["value"]))
except Exception as e:
    print("$n", "FAILED", e)
PY
}
run full_l2_f16 --lane-cells 2 --fuse 16
for fuse in 12 14 16 18 20; do run full_l22_f$fuse --lane-cells 22 --fuse $fuse; done
for fuse in 8 12 14 16; do run slab1024_l22_f$fuse --dim-y 1024 --lane-cells 22 --fuse $fuse; done
for fuse in 12 16; do run slab2048_l22_f$fuse --dim-y 2048 --lane-cells 22 --fuse $fuse; done
