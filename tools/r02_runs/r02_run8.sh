#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run8
mkdir -p $O
./tools/ubench_pk_chain > $O/ubench_pk_chain.log 2>&1; cat $O/ubench_pk_chain.log
( time python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; head -c 1500 $O/bench_default.json; echo
