#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_final4; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | tail -1 | tee $O/smoke.log
timeout 500 python tools/soak_overlap.py 71 200 2>&1 | tail -3 | tee $O/soak_steps.txt
