#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "step_n or step" ) > $O/pytest_gpu_stepn.log 2>&1
tail -12 $O/pytest_gpu_stepn.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sim-steps 8 > $O/bench_stepn.json 2> $O/bench_stepn.err || tail -3 $O/bench_stepn.err
python -c "
import json;d=json.load(open('$O/bench_stepn.json'));print('solve %.4f ms  sim steps/s %.1f (step_n / best)  %.1f (separate calls)' % (d['ms_per_step'], d['sim_steps_per_sec'], d['sim_steps_per_sec_as_separate_calls']))"
