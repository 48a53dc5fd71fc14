#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run11
mkdir -p $O
( time python -m pytest tests/test_gpu_parity.py tests/test_dropin_headers.py -m gpu -x -q -k "operators or golden or steps or full_step or irregular or dropin or context_loop or config2 or long_run" ) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^E " $O/pytest_gpu.log | head -5
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02_run11/bench_default.json"))
print(d["value"], d["ms_per_step"], d["sim_steps_per_sec"], d["parity"]["bit_exact"])
for k,v in d["sim_step_per_operator"].items(): print(k, round(v["us"],1), round(v.get("frac_of_hbm_peak",0),3))
PY
