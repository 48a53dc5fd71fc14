#!/bin/bash
# r04: fuzz of the single-GPU path (solve, 2-3 steps through sfl_step_n with the seam kernel, a generic advection per configuration) and
# soak of the overlapped slab executor with the device-side arrival count, against the oracle / a whole-domain context
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
( python tests/fuzz_vs_oracle.py 41 ${1:-300}; python tests/fuzz_vs_oracle.py 42 ${2:-200} big; python tools/soak_overlap.py 43 ${3:-240} ) 2>&1 | grep -v "^$" | tee $O/soak_fuzz.txt | tail -12
