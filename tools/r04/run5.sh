#!/bin/bash
# r04: device-side halo arrival (SFL_OPT_SOR_ARRIVAL): parity on virtual ranks, then the emulated rank with and without it
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "overlapped or arrival or virtual or config4 or config5 or emulated or slab" ) > $O/pytest_gpu_arrival.log 2>&1
tail -6 $O/pytest_gpu_arrival.log
for rep in 1 2; do
for arr in 1 0; do
  for r in 3 0; do
    timeout 300 python bench.py $([ $arr = 0 ] && echo --arrival-by-event) --emulate-rank $r --of 8 --steps 20 --warmup 3 --sim-steps 6 > $O/emu_arr.json 2> $O/emu_arr.err || tail -3 $O/emu_arr.err
    python -c "import json;d=json.load(open('$O/emu_arr.json'));print('arrival by flag $arr  rank $r of 8: %.4f ms per solve, sim step %.1f us' % (d['ms_per_solve'], d['sim_step_us'] or 0))" | tee -a $O/arrival_ab.txt
  done
done; done
