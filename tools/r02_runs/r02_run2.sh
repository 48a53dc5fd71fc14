#!/bin/bash
# Round 2, GPU call 2: twin tiles -- parity tests, then timing sweeps (lane 2 vs 22) on the whole
# domain and on the slab shares.
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run2
mkdir -p $O
( time python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "twin or fused_vs_oracle or randomised or spot_check or config3 or config2 or virtual_slabs or large_grid" ) > $O/pytest_twin.log 2>&1
tail -5 $O/pytest_twin.log
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
for lane in 2 22; do
  for fuse in 12 16; do run full_l${lane}_f${fuse} --lane-cells $lane --fuse $fuse; done
  for dy in 1024 2048; do
    for fuse in 8 12 16; do run slab${dy}_l${lane}_f${fuse} --dim-y $dy --lane-cells $lane --fuse $fuse; done
  done
done
# rows-per-tile sweep of the twin flavour
for rows in 12 16 20 26 34 44; do run slab1024_l22_f12_r$rows --dim-y 1024 --lane-cells 22 --fuse 12 --sor-rows $rows; done
for rows in 40 60 80 120 160; do run full_l22_f16_r$rows --lane-cells 22 --fuse 16 --sor-rows $rows; done
