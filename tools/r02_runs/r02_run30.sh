#!/bin/bash
# advection tiles in panels of 16 tile columns: parity, step trace and HBM-side counters
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run30
mkdir -p $O
( python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "advection_kernels or fused_advection or host_advect or golden or forces or step or irregular" ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -1; grep -E "^E " $O/pytest.log | head -5
bash profiles/run_step_trace.sh > gpurun_out/r02_run30_step_trace.txt 2>&1
bash profiles/run_step_pmc.sh > gpurun_out/r02_run30_step_pmc.txt 2>&1
head -16 gpurun_out/r02_run30_step_trace.txt
grep -E "FETCH_SIZE" gpurun_out/r02_run30_step_pmc.txt | grep tiled
