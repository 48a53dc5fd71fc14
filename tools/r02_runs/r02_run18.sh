#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run18
mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -1; grep -E "^E " $O/pytest_gpu.log | head -3
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02_run18/bench_default.json"))
print(d["value"], d["value_unprimed"], d["ms_per_step"], d["sim_steps_per_sec"], d["parity"]["bit_exact"], d["roofline"]["frac"], d["roofline"]["valu"]["frac"])
PY
B="python bench.py --no-cpu-baseline --sim-steps 0 --steps 30 --warmup 5"
for a in "--dim-y 4096" "--dim-y 2048" "--dim-y 1024" "--size 16384 --iters 200 --steps 5 --warmup 2"; do $B $a 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$a', d['ms_per_step'], d['config']['half_sweeps_fused_per_launch'])"; done
bash profiles/run_profile.sh r02_final > $O/profile.log 2>&1; grep -E "NS=16, dx1=true, zero_in=false" gpurun_out/prof_r02_final/summary.txt | head -30
