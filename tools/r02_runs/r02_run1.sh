#!/bin/bash
# Round 2, GPU call 1: full GPU test suite, default bench, slab-share timings, slab PMC profile.
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run1
mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
cat $O/bench_default.json | head -c 3000
for dy in 1024 2048 4096; do
  python bench.py --dim-y $dy --no-cpu-baseline --sim-steps 0 --steps 20 --warmup 5 > $O/bench_slab_$dy.json 2>> $O/bench_slab.err
  python - <<PY
import json
d=json.load(open("$O/bench_slab_$dy.json"))
print("slab $dy", d["ms_per_step"], d["config"]["half_sweeps_fused_per_launch"], d["config"]["sor_launches_per_solve"], d["roofline"]["avg_launch_us"])
PY
done
bash profiles/run_profile.sh r02_slab1024 --dim-y 1024 > $O/profile_slab1024.log 2>&1
tail -30 $O/profile_slab1024.log
