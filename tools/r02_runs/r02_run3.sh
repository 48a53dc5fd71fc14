#!/bin/bash
# Round 2, GPU call 3: twin tiles after the DPP-hazard fix -- parity tests, timings, PMC of twin vs scalar on a slab.
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run3
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
  run full_l${lane}_f16 --lane-cells $lane --fuse 16
  for fuse in 8 12 16; do run slab1024_l${lane}_f${fuse} --dim-y 1024 --lane-cells $lane --fuse $fuse; done
done
for lane in 2 22; do
  bash profiles/run_pmc_custom.sh r02_slab1024_l${lane}_f12 --dim-y 1024 --lane-cells $lane --fuse 12 > $O/pmc_l${lane}.log 2>&1
  cp gpurun_out/prof_r02_slab1024_l${lane}_f12/summary.txt $O/pmc_summary_l${lane}.txt
  grep -E "NS=12, dx1=true, zero_in=false" $O/pmc_summary_l${lane}.txt
done
