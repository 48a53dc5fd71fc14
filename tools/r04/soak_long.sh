#!/bin/bash
# r04: long soak of the library as committed at the end of the round (one box, ~17 minutes)
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_final4; mkdir -p $O
( python tools/soak_overlap.py 81 420; python tests/fuzz_vs_oracle.py 82 240; python tools/r04/chain_stress.py 180 83; python tools/r04/unaligned_stress.py 120 1 ) 2>&1 | grep -v "^$" | tee $O/soak_long.txt | tail -8
