#!/bin/bash
# final in-time executor: full GPU suite, stress on unaligned pitches (both schemes), soak, emulated-rank A/B
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu_full2.log 2>&1
grep -E "passed|failed" $O/pytest_gpu_full2.log
for a in 1 0; do timeout 600 python tools/r04/unaligned_stress.py 240 $a 2>&1 | tail -4 | tee -a $O/unaligned_stress_final.txt; done
timeout 900 python tools/soak_overlap.py 44 300 2>&1 | grep -v "^$" | tail -5 | tee $O/soak_final.txt
for rep in 1 2 3; do
for mode in "" "--arrival-by-event"; do
  for r in 3 0; do
    timeout 300 python bench.py $mode --emulate-rank $r --of 8 --steps 30 --warmup 5 --sim-steps 8 > $O/emu_it.json 2> $O/emu_it.err || tail -3 $O/emu_it.err
    python -c "import json;d=json.load(open('$O/emu_it.json'));print('%-34s rank $r of 8: %.4f ms per solve, sim step %.1f us, %d exchanges' % ('${mode:-in time, counted on the device}', d['ms_per_solve'], d['sim_step_us'] or 0, d['halo_exchanges_per_solve']))" | tee -a $O/intime_ab_final.txt
  done
done; done
