#!/bin/bash
# r04: the chained launch across in-time exchanges on the emulated rank (timing only), against single launches; same box
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chained_launch or emulated or overlapped" ) 2>&1 | tail -3
for rep in 1 2 3; do
for ch in 0 1; do
  timeout 300 python bench.py --emulate-rank 3 --of 8 --steps 30 --warmup 5 --sim-steps 8 --chain $ch > $O/emu_chain.json 2> $O/emu_chain.err || tail -3 $O/emu_chain.err
  python -c "import json;d=json.load(open('$O/emu_chain.json'));print('rank 3 of 8 chain $ch: %.4f ms per solve, sim step %.1f us, %d launches' % (d['ms_per_solve'], d['sim_step_us'] or 0, d['sor_launches_per_solve']))" | tee -a $O/emu_chain_ab.txt
done; done
for ch in 0 1; do
  timeout 300 python bench.py --dim-y 1024 --no-cpu-baseline --sim-steps 0 --steps 30 --warmup 5 --chain $ch > $O/chain.json 2> $O/chain.err || tail -3 $O/chain.err
  python -c "import json;d=json.load(open('$O/chain.json'));print('8192x1024 chain $ch: %.4f ms per solve' % d['ms_per_step'])" | tee -a $O/emu_chain_ab.txt
done
