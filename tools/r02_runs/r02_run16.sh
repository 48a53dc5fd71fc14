#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run16
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
run base_full
for v in st2 ld2 all2 st17; do SFL_LIB=$PWD/tools/variants/libsfl_$v.so run ${v}_full; done
run base_full_again
for v in st2 all2; do SFL_LIB=$PWD/tools/variants/libsfl_$v.so run ${v}_c5 --size 16384 --iters 200 --steps 5 --warmup 2; done
run base_c5 --size 16384 --iters 200 --steps 5 --warmup 2
