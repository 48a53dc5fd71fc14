#!/bin/bash
# r04: validation of the final tree on one box: the whole GPU suite, smoke(), the default bench line and the two small configurations
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_validate
mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | tail -2 | tee $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err || tail -3 $O/bench_default.err
python bench.py --size 2048 --iters 40 > $O/bench_c2_2048.json 2> $O/bench_c2.err
python bench.py --size 61 --dim-y 81 --iters 20 > $O/bench_c1_61x81.json 2> $O/bench_c1.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_validate/bench_default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'])
print(d['sim_steps_per_sec'], d.get('sim_step_kernels'))
PY
