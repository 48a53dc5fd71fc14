#!/bin/bash
# ON THE GPU BOX ("measure first"): what would a step gain if the pressure never went through memory between the
# solve's last launch and the projection inside sfl_step_n?  Two TIMING MOCKS (wrong results, diagnostic builds only):
#   build/probes/sor_clock_probe_ns16_{store,nostore}   the NS = 16 launch at 8192^2 with and without its stores of p
#   build/variants/seam_nop/libsfl_hip.so               the seam kernel without its loads of the pressure window
# built by: bash tools/recipes/build_variant.sh probe ns16_store 16 "" ; ... probe ns16_nostore 16 "-DSFL_PROBE_NO_STORE=1" ;
#           ... lib seam_nop "-DSEAM_MOCK_NO_P=1 -DSFL_ALLOW_TIMING_MOCKS" advect_tiled.hip
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_p_never_stored.txt; : > $O
for rep in 1 2 3; do for v in store nostore; do
  echo "rep $rep $v: $(build/probes/sor_clock_probe_ns16_$v 8192 8192 40 2>&1 | grep -m1 'launch by HIP events' | sed 's/.*launch by/launch by/')" | tee -a $O
done; done
for rep in 1 2; do for v in product seam_nop; do
  if [ $v = product ]; then unset LIB; else export LIB=build/variants/seam_nop/libsfl_hip.so; fi
  SIM_STEPS=20 bash profiles/run_step_pmc.sh r06_pns_$v > /dev/null 2>&1
  echo "rep $rep $v: $(grep -m1 '^seam_tiled_kernel' gpurun_out/prof_step_r06_pns_$v/summary.txt | cut -c96-230)" | tee -a $O
  grep -E "^seam_tiled_kernel +FETCH" gpurun_out/prof_step_r06_pns_$v/summary.txt | cut -c1-200 | tee -a $O
done; done
unset LIB
for rep in 1 2 3; do for v in product seam_nop; do
  if [ $v = product ]; then run="python bench.py"; else run="python tools/with_lib.py build/variants/seam_nop/libsfl_hip.so bench.py"; fi
  $run --steps 3 --warmup 1 --no-cpu-baseline --no-fold-leg --sim-steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v rep $rep: %.1f steps/s (step_n)  %.1f us per step' % (d['sim_steps_per_sec'], d['sim_step_us']))" | tee -a $O
done; done
