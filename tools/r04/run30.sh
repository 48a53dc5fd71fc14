#!/bin/bash
# r04: wait + copy + signal as one kernel (launch_halo_copy): slab parity tests, then the emulated ranks with single and chained launches
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
true
for rep in 1 2 3; do
for ch in 0 1; do
  timeout 300 python bench.py --emulate-rank 3 --of 8 --steps 30 --warmup 5 --sim-steps 8 --chain $ch > $O/emu_chain.json 2> $O/emu_chain.err || tail -3 $O/emu_chain.err
  python -c "import json;d=json.load(open('$O/emu_chain.json'));print('halo_copy: rank 3 of 8 chain $ch: %.4f ms per solve, sim step %.1f us' % (d['ms_per_solve'], d['sim_step_us'] or 0))" | tee -a $O/halo_copy_ab.txt
done; done
bash tools/r04/trace_emulate.sh halo_copy --emulate-rank 3 --of 8 | sed -n 28,45p
