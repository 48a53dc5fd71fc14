#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_final3
mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | tail -1 | tee $O/smoke.log
bash tools/r04/final_evidence.sh > $O/final_evidence.log 2>&1
cat $O/emulate_c4.txt $O/emulate_c5.txt
