#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run15
mkdir -p $O
( python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_vs_oracle or randomised or config2 or spot_check or virtual_slabs_match or overlapped" ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -1; grep -E "^E " $O/pytest.log | head -3
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
run full_auto
run full_f12 --fuse 12
run full_f14 --fuse 14
for fuse in 8 10 12 14 16; do run slab1024_f$fuse --dim-y 1024 --fuse $fuse; done
for rows in 48 64 80 100; do run slab1024_f10_r$rows --dim-y 1024 --fuse 10 --sor-rows $rows; run slab1024_f12_r$rows --dim-y 1024 --fuse 12 --sor-rows $rows; done
run slab2048_auto --dim-y 2048
run slab2048_f12 --dim-y 2048 --fuse 12
run slab4096_auto --dim-y 4096
run c2_2048 --size 2048 --iters 40
