#!/bin/bash
# r04: the chained launch (SFL_OPT_SOR_CHAIN): parity, then A/B on the thin share and on the headline grid
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chained_launch" ) 2>&1 | tail -5
for rep in 1 2; do
for ch in 0 1; do
  timeout 300 python bench.py --dim-y 1024 --no-cpu-baseline --sim-steps 0 --steps 30 --warmup 5 --chain $ch > $O/chain.json 2> $O/chain.err || tail -3 $O/chain.err
  python -c "import json;d=json.load(open('$O/chain.json'));print('8192x1024 chain $ch: %.4f ms per solve' % d['ms_per_step'])" | tee -a $O/chain_ab.txt
done; done
for ch in 0 1; do
  timeout 300 python bench.py --no-cpu-baseline --sim-steps 0 --chain $ch > $O/chain.json 2> $O/chain.err || tail -3 $O/chain.err
  python -c "import json;d=json.load(open('$O/chain.json'));print('8192x8192 chain $ch: %.4f ms per solve' % d['ms_per_step'])" | tee -a $O/chain_ab.txt
done
