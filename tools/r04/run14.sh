#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q -k "not (config5 or config4 or headline or large or eight)" ) > $O/pytest_gpu_async.log 2>&1
tail -6 $O/pytest_gpu_async.log
for rep in 1 2 3; do
for a in 1 0; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sim-steps 10 $([ $a = 0 ] && echo --no-async-dye) > $O/bench_async.json 2> $O/bench_async.err || tail -3 $O/bench_async.err
  python -c "
import json;d=json.load(open('$O/bench_async.json'));print('async dye $a: solve %.4f ms  sim steps/s %.1f  (%.1f us per step)' % (d['ms_per_step'], d['sim_steps_per_sec'], d['sim_step_us']))" | tee -a $O/async_dye_ab.txt
done; done
