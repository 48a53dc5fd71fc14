#!/bin/bash
# ON THE GPU BOX: the sim step's kernels on variant builds of the library (tools/recipes/build_variant.sh lib <name> ... advect_tiled.hip),
# "product" = the library as built: per variant a kernel trace + FETCH / WRITE passes over SIM_STEPS sim steps (steady-state rows of the
# three step kernels), then interleaved steps/s.     gpurun -- 'bash tools/recipes/step_variant_sweep.sh <tag> product <name> [<name> ...]'
set -u
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/stepsweep_$TAG.txt; : > $OUT
for v in "$@"; do
  if [ $v = product ]; then unset LIB; else export LIB=build/variants/$v/libsfl_hip.so; fi
  SIM_STEPS=${SIM_STEPS:-20} bash profiles/run_step_pmc.sh sw_$v > /dev/null 2>&1
  echo "== $v" >> $OUT
  grep -E "^seam_tiled_kernel|^advect_divergence_tiled|^advect_vec3uq32_tiled_kernel<no_slip=false, fuse_grad=true" gpurun_out/prof_step_sw_$v/summary.txt \
    | grep -E "FETCH| [0-9]+ +[0-9.]+ +[0-9.]+ " | cut -c1-215 >> $OUT
done
unset LIB
for rep in 1 2 3; do for v in "$@"; do
  if [ $v = product ]; then run="python bench.py"; else run="python tools/with_lib.py build/variants/$v/libsfl_hip.so bench.py"; fi
  $run --steps 3 --warmup 1 --no-cpu-baseline --no-fold-leg --sim-steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v rep $rep: %.1f steps/s (step_n)  %.1f (separate calls)' % (d['sim_steps_per_sec'], d['sim_steps_per_sec_as_separate_calls']))" >> $OUT
done; done
cat $OUT
