#!/bin/bash
# r04: soak / fuzz / stress of the FINAL library, one box: fuzz against the oracle, the overlapped slab executor on virtual ranks,
# unaligned pitches in both arrival modes, the chained launch (whole domains, 2 / 3 virtual ranks)
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_final3
mkdir -p $O
( python tests/fuzz_vs_oracle.py 51 240; python tests/fuzz_vs_oracle.py 52 120 big; python tools/soak_overlap.py 53 240; \
  python tools/r04/unaligned_stress.py 150 1; python tools/r04/unaligned_stress.py 100 0; python tools/r04/chain_stress.py 200 17 ) 2>&1 \
  | grep -v "^$" | tee $O/soak_final.txt | tail -14
