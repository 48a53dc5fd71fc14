#!/bin/bash
# r04 second GPU call: the generic-advection tests, the per-wave trace of the thin share, the emulated rank's sim step
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
( time python -m pytest tests -m gpu -x -q -k "generic or external or dropin or domain_for_each" ) > $O/pytest_gpu_generic.log 2>&1
tail -5 $O/pytest_gpu_generic.log
./tools/sor_clock_probe_ns10 8192 1024 40 0 $O/probe_ns10_slab1024.csv > $O/probe_ns10_slab1024.txt 2>&1
cat $O/probe_ns10_slab1024.txt
for D in 0 25 50; do
  python bench.py --emulate-rank 3 --of 8 --wire-us $D --steps 20 --warmup 3 --sim-steps 6 > $O/emu_step_$D.json 2> $O/emu_step.err || tail -3 $O/emu_step.err
  python -c "import json;d=json.load(open('$O/emu_step_$D.json'));print('rank 3 of 8, D=$D: %.4f ms per solve, sim step %s us %s' % (d['ms_per_solve'], d['sim_step_us'], d.get('sim_steps_note','')))"
done
