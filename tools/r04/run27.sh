#!/bin/bash
# r04: where the chained launch loses: variants without the waits / with plain loads and stores (wrong results), and a kernel trace
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
for v in product chain_stride1 chain_sleep8 chain_nodeps; do
  if [ $v = product ]; then run="python bench.py"; else run="python tools/with_lib.py build/variants/$v/libsfl_hip.so bench.py"; fi
  timeout 300 $run --dim-y 1024 --no-cpu-baseline --no-parity --sim-steps 0 --steps 30 --warmup 5 --chain 1 > $O/chain.json 2> $O/chain.err || tail -3 $O/chain.err
  python -c "import json;d=json.load(open('$O/chain.json'));print('8192x1024 chain 1 %-20s: %.4f ms per solve' % ('$v', d['ms_per_step']))" | tee -a $O/chain_variants.txt
done
