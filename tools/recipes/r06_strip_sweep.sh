#!/bin/bash
# ON THE GPU BOX: the tiled step kernels' tile order (SFL_STRIP_TILES: 0 = row-major as in rounds 2-5, 8 / 16 / 32 tile columns per strip):
# kernel trace + FETCH / WRITE passes over 20 sim steps per variant, interleaved steps/s.   gpurun -- bash tools/recipes/r06_strip_sweep.sh
set -u
export TMPDIR=/tmp
for v in ${STRIPS:-0 16 8 32}; do
  if [ $v = 16 ]; then unset LIB; else export LIB=build/variants/strip$v/libsfl_hip.so; fi
  SIM_STEPS=20 bash profiles/run_step_pmc.sh r06_strip$v > /dev/null 2>&1
  grep -E "seam_tiled_kernel|advect_divergence_tiled|advect_vec3uq32_tiled_kernel<no_slip=false, fuse_grad=true" gpurun_out/prof_step_r06_strip$v/summary.txt | cut -c1-230 > gpurun_out/r06_strip${v}_rows.txt
done
unset LIB
: > gpurun_out/r06_strip_steps.txt
for rep in 1 2 3; do for v in ${STRIPS:-0 16 8 32}; do
  if [ $v = 16 ]; then run="python bench.py"; else run="python tools/with_lib.py build/variants/strip$v/libsfl_hip.so bench.py"; fi
  $run --steps 3 --warmup 1 --no-cpu-baseline --no-fold-leg --sim-steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('strip $v rep $rep: %.1f steps/s (step_n)  %.1f (separate calls)' % (d['sim_steps_per_sec'], d['sim_steps_per_sec_as_separate_calls']))" | tee -a gpurun_out/r06_strip_steps.txt
done; done
for v in ${STRIPS:-0 16 8 32}; do echo "== strip $v"; cat gpurun_out/r06_strip${v}_rows.txt; done
