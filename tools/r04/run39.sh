#!/bin/bash
set -u
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "early_interior" ) 2>&1 | grep -E "^E  |passed|failed" | head
