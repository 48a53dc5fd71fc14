#!/bin/bash
set -u
export TMPDIR=/tmp
for rep in 1 2 3; do
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "chained_launch_across" ) 2>&1 | grep -E "^E  |passed|failed" | head -30
done
( GPU_MAX_HW_QUEUES=4 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "chained_launch_across" ) 2>&1 | grep -E "^E  |passed|failed" | head -30
