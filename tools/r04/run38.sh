#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_final3; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "slab_steps or automatic_advection or config4_whole or overlapped_exchange_inside or emulated or both_advection_kernels_on_slabs or step" ) 2>&1 | grep -E "^E  |passed|failed" | head
for rep in 1 2 3; do
  python bench.py --emulate-rank $((rep*3-3)) --of 8 --steps 10 --warmup 3 --sim-steps 12 > $O/run.json 2>$O/run.err || tail -3 $O/run.err
  python -c "import json;d=json.load(open('$O/run.json'));print('rank 3 of 8: %.4f ms per solve, sim step %.1f us' % (d['ms_per_solve'], d['sim_step_us'] or 0))"
done
timeout 400 python tools/soak_overlap.py 62 150 2>&1 | tail -3
