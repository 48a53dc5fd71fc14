#!/bin/bash
# profile refresh: sim-step kernel trace + HBM-side counters of the step kernels, default bench line
set -u
export TMPDIR=/tmp
bash profiles/run_step_trace.sh > gpurun_out/r02_run29_step_trace.txt 2>&1
bash profiles/run_step_pmc.sh > gpurun_out/r02_run29_step_pmc.txt 2>&1
python bench.py > gpurun_out/r02_run29_bench.json 2> gpurun_out/r02_run29_bench.err
tail -c 3000 gpurun_out/r02_run29_bench.json
head -40 gpurun_out/r02_run29_step_trace.txt
