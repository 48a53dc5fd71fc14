#!/bin/bash
# rhs read-ahead depth x prefetch rows: library variants A/B
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run36
mkdir -p $O
B="python bench.py --no-cpu-baseline --sim-steps 0 --steps 30 --warmup 5"
run() { n=$1; shift; "$@" > $O/$n.json 2>> $O/bench.err; python - <<PY
import json
try:
    d=json.load(open("$O/$n.json"))
    print("%-16s %8.4f ms  fuse %2d launches %2d  %7.2f us/launch" % ("$n", d["ms_per_step"], d["config"]["half_sweeps_fused_per_launch"], d["config"]["sor_launches_per_solve"], d["roofline"]["avg_launch_us"]))
except Exception as e:
    print("$n", "FAILED", e)
PY
}
for rep in 1 2; do
for cfg in "full:" "s2048:--dim-y 2048" "s1024:--dim-y 1024" "c2:--size 2048 --iters 40" "c5s:--size 16384 --dim-y 2048 --iters 200 --steps 8"; do
  n=${cfg%%:*}; a=${cfg#*:}
  run ${n}_base_$rep $B $a
  for v in a1p6 a2p6 a2p3 a1p3 a3p3 a0p3; do
    SFL_LIB=$PWD/esp32-fluid-simulation_amd/lib/variants/libsfl_hip_$v.so run ${n}_${v}_$rep $B $a
  done
done; done
