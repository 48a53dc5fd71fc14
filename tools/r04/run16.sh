#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_dropin_headers.py -m gpu -x -q -k "step" ) > $O/pytest_gpu_stepn.log 2>&1
tail -3 $O/pytest_gpu_stepn.log
for rep in 1 2 3; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sim-steps 10 > $O/bench_stepn.json 2> $O/bench_stepn.err || tail -3 $O/bench_stepn.err
  python -c "
import json;d=json.load(open('$O/bench_stepn.json'));print('sfl_step_n %.1f steps/s (%.1f us per step)   n x sfl_step %.1f steps/s' % (d['sim_steps_per_sec'], d['sim_step_us'], d['sim_steps_per_sec_as_separate_calls']))" | tee -a $O/stepn_ab2.txt
done
