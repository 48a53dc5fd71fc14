#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run39
mkdir -p $O
( python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_vs_oracle or randomised or config2 or config3 or spot_check or virtual_slabs or overlapped or irregular" ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -1; grep -E "^E " $O/pytest.log | head -5
for a in "--fuse 14" "--fuse 12" "" "--dim-y 1024" "--dim-y 1024 --fuse 8"; do
python bench.py --no-cpu-baseline --sim-steps 0 --steps 30 --warmup 5 $a > $O/b.json 2>> $O/err.log
python - <<PY
import json
d = json.load(open("$O/b.json"))
print("%-22s %.4f ms  fuse %2d launches %2d  %.2f us/launch" % ("$a", d["ms_per_step"], d["config"]["half_sweeps_fused_per_launch"], d["config"]["sor_launches_per_solve"], d["roofline"]["avg_launch_us"]))
PY
done
