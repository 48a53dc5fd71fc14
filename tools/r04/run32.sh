#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
timeout 400 python tools/r04/chain_stress.py 240 2>&1 | tail -8 | tee $O/chain_stress.txt
( timeout 2400 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu_chain.log 2>&1; grep -E "passed|failed" $O/pytest_gpu_chain.log | tail -2
