#!/bin/bash
# full GPU suite + smoke + default bench line on the current library
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu_full.log 2>&1
grep -E "passed|failed" $O/pytest_gpu_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/bench_default2.json 2> $O/bench_default2.err || tail -3 $O/bench_default2.err
python -c "
import json;d=json.load(open('$O/bench_default2.json'))
print('value %.4e  ms/step %.4f  unprimed %.4e  sim steps/s %.1f (separate calls %.1f)  parity %s  frac %.3f  traffic_source %s' % (d['value'], d['ms_per_step'], d['value_unprimed'], d['sim_steps_per_sec'], d['sim_steps_per_sec_as_separate_calls'], d['parity']['bit_exact'], d['roofline']['frac'], d['roofline']['traffic_source']))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['kind'])
print(str(d['sim_step_kernels'])[:200])"
