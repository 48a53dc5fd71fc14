#!/bin/bash
# the round's numbers on the final library: tests, evidence, wire curve, refit sweep
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu_full3.log 2>&1
grep -E "passed|failed" $O/pytest_gpu_full3.log
bash tools/r04/final_evidence.sh > $O/final_evidence.log 2>&1
tail -4 $O/final_evidence.log
rm -f $O/wire_curve.txt; bash tools/r04/wire_curve.sh > /dev/null 2>&1; cat $O/wire_curve.txt
rm -f $O/refit_sweep.txt; bash tools/r04/refit_sweep.sh > /dev/null 2>&1; cat $O/refit_sweep.txt
