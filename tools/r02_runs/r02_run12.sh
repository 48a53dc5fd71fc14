#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run12
mkdir -p $O
( time python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config5 or group_options" ) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^E " $O/pytest_gpu.log | head -5
B="python bench.py --no-cpu-baseline --sim-steps 0 --steps 20 --warmup 5"
run() { n=$1; shift; $B "$@" > $O/$n.json 2>> $O/bench.err; python - <<PY
import json
try:
    d=json.load(open("$O/$n.json"))
    print("%-30s %8.4f ms  fuse %2d launches %2d  %7.2f us/launch  %.3e  unprimed %.3e" % ("$n", d["ms_per_step"], d["config"]["half_sweeps_fused_per_launch"], d["config"]["sor_launches_per_solve"], d["roofline"]["avg_launch_us"], d["value"], d["value_unprimed"] or 0))
except Exception as e:
    print("$n", "FAILED", e)
PY
}
run full_8192
run share2_8192x4096 --dim-y 4096
run share4_8192x2048 --dim-y 2048
run share8_8192x1024 --dim-y 1024
run c2_2048 --size 2048 --iters 40
run c1_61x81 --size 61 --dim-y 81 --iters 20
run c5_16384_1gpu --size 16384 --iters 200 --steps 5 --warmup 2
run c5_share8_16384x2048 --size 16384 --dim-y 2048 --iters 200 --steps 10 --warmup 3
