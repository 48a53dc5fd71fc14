#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_final3
mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err || tail -3 $O/bench_default.err
python bench.py --size 2048 --iters 40 > $O/bench_c2_2048.json 2> $O/bench_c2.err
python bench.py --size 61 --dim-y 81 --iters 20 > $O/bench_c1_61x81.json 2> $O/bench_c1.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_final3/bench_default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'])
print(d['sim_steps_per_sec'], list((d.get('sim_step_kernels') or {}).keys()) if isinstance(d.get('sim_step_kernels'), dict) else d.get('sim_step_kernels'))
PY
