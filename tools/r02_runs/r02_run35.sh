#!/bin/bash
# rhs values read from the LDS ring one stage ahead of their use: A/B
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run35
mkdir -p $O
V=$PWD/esp32-fluid-simulation_amd/lib/variants/libsfl_hip_ahead.so
( SFL_LIB=$V python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_vs_oracle or randomised or config2 or config3 or spot_check or virtual_slabs or overlapped or irregular" ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -1; grep -E "^E " $O/pytest.log | head -5
B="python bench.py --no-cpu-baseline --sim-steps 0 --steps 30 --warmup 5"
run() { n=$1; shift; "$@" > $O/$n.json 2>> $O/bench.err; python - <<PY
import json
try:
    d=json.load(open("$O/$n.json"))
    print("%-22s %8.4f ms  fuse %2d launches %2d  %7.2f us/launch" % ("$n", d["ms_per_step"], d["config"]["half_sweeps_fused_per_launch"], d["config"]["sor_launches_per_solve"], d["roofline"]["avg_launch_us"]))
except Exception as e:
    print("$n", "FAILED", e)
PY
}
for rep in 1 2; do
for cfg in "full:" "s4096:--dim-y 4096" "s2048:--dim-y 2048" "s1024:--dim-y 1024" "c2:--size 2048 --iters 40" "c5s:--size 16384 --dim-y 2048 --iters 200 --steps 8" "full12:--fuse 12" "full8:--fuse 8"; do
  n=${cfg%%:*}; a=${cfg#*:}
  run base_${n}_$rep $B $a
  SFL_LIB=$V run ahead_${n}_$rep $B $a
done; done
