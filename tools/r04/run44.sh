#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_final4; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "slab or automatic_advection or config4_whole or overlapped_exchange_inside or emulated or advection or step or early_interior or virtual" ) 2>&1 | grep -E "^E  |passed|failed" | head
for rep in 1 2 3; do
for cfg in "--emulate-rank 3 --of 8" "--emulate-rank 1 --of 4"; do
  python bench.py --steps 10 --warmup 3 --sim-steps 12 $cfg > $O/run.json 2>$O/run.err || tail -3 $O/run.err
  python -c "import json;d=json.load(open('$O/run.json'));print('$cfg: %.4f ms per solve, sim step %.1f us' % (d['ms_per_solve'], d['sim_step_us'] or 0))"
done; done
timeout 300 python tools/soak_overlap.py 91 100 2>&1 | tail -2
bash profiles/run_step_pmc.sh r04 > $O/profile_step.log 2>&1; tail -2 $O/profile_step.log
