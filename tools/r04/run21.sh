#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
for rep in 1 2 3; do
for mode in "" "--early-exchanges --arrival-by-event"; do
    timeout 300 python bench.py $mode --emulate-rank 3 --of 8 --steps 30 --warmup 5 --sim-steps 8 > $O/emu_it.json 2> $O/emu_it.err || tail -3 $O/emu_it.err
    python -c "import json;d=json.load(open('$O/emu_it.json'));print('%-40s rank 3 of 8: %.4f ms per solve, sim step %.1f us, %d exchanges' % ('${mode:-in time (sender tiles, written-through)}', d['ms_per_solve'], d['sim_step_us'] or 0, d['halo_exchanges_per_solve']))" | tee -a $O/intime_ab2.txt
done; done
for D in 0 10 25 50; do
  for mode in "" "--early-exchanges --arrival-by-event"; do
    timeout 300 python bench.py $mode --emulate-rank 3 --of 8 --wire-us $D --steps 20 --warmup 3 --sim-steps 0 > $O/emu_it.json 2> $O/emu_it.err || tail -3 $O/emu_it.err
    python -c "import json;d=json.load(open('$O/emu_it.json'));print('D=%3d us  %-36s %.4f ms per solve' % ($D, '${mode:-in time}', d['ms_per_solve']))" | tee -a $O/intime_wire2.txt
  done
done
bash tools/r04/trace_emulate.sh intime2 --emulate-rank 3 --of 8 | sed -n 12,36p
timeout 400 python tools/r04/unaligned_stress.py 150 1 1 2>&1 | tail -3
