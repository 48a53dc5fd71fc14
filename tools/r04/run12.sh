#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "step_n" ) > $O/pytest_gpu_stepn.log 2>&1
tail -4 $O/pytest_gpu_stepn.log
for rep in 1 2; do
for v in product seam_w4; do
  if [ $v = product ]; then run="python bench.py"; else run="python tools/with_lib.py build/variants/$v/libsfl_hip.so bench.py"; fi
  $run --steps 10 --warmup 3 --no-cpu-baseline --sim-steps 8 > $O/bench_stepn.json 2> $O/bench_stepn.err || tail -3 $O/bench_stepn.err
  python -c "
import json;d=json.load(open('$O/bench_stepn.json'));print('$v: solve %.4f ms  sim steps/s %.1f (best of step_n and separate)  %.1f (separate calls)' % (d['ms_per_step'], d['sim_steps_per_sec'], d['sim_steps_per_sec_as_separate_calls']))" | tee -a $O/stepn_ab.txt
done; done
bash tools/r04/run11.sh
