#!/bin/bash
# r04: chained launch on the headline grid and on mid sizes (padded flag words), A/B on one box
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
for cfg in "--size 8192" "--size 8192 --dim-y 4096" "--size 8192 --dim-y 2048" "--size 4096 --iters 40" "--size 2048 --iters 40"; do
for rep in 1 2; do
for ch in 0 1; do
  timeout 300 python bench.py $cfg --no-cpu-baseline --sim-steps 0 --chain $ch > $O/chain.json 2> $O/chain.err || tail -3 $O/chain.err
  python -c "import json;d=json.load(open('$O/chain.json'));print('%-28s chain $ch: %.4f ms per solve' % ('$cfg', d['ms_per_step']))" | tee -a $O/chain_sizes.txt
done; done; done
