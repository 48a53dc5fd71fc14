#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run17
mkdir -p $O
B="python bench.py --no-cpu-baseline --sim-steps 0 --steps 30 --warmup 5"
run() { n=$1; shift; $B "$@" > $O/$n.json 2>> $O/bench.err; python - <<PY
import json
try:
    d=json.load(open("$O/$n.json"))
    print("%-30s %8.4f ms  fuse %2d launches %2d  %7.2f us/launch" % ("$n", d["ms_per_step"], d["config"]["half_sweeps_fused_per_launch"], d["config"]["sor_launches_per_solve"], d["roofline"]["avg_launch_us"]))
except Exception as e:
    print("$n", "FAILED", e)
PY
}
for rep in 1 2; do
for cfg in "full:" "s4096:--dim-y 4096" "s2048:--dim-y 2048" "s1024:--dim-y 1024" "c2:--size 2048 --iters 40"; do
  n=${cfg%%:*}; a=${cfg#*:}
  run base_${n}_$rep $a
  SFL_LIB=$PWD/tools/variants/libsfl_st2_all.so run nt_${n}_$rep $a
done; done
