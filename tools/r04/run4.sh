#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
( time python -m pytest tests/test_dropin_headers.py -m gpu -x -q ) > $O/pytest_gpu_headers.log 2>&1
tail -15 $O/pytest_gpu_headers.log
