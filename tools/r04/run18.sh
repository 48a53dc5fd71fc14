#!/bin/bash
# r04: in-time exchanges with sender tiles (SFL_OPT_SOR_IN_TIME) against early exchanges: parity on virtual ranks, emulated rank A/B, wire curve, timeline
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "overlapped or arrival or virtual or config4 or config5 or emulated or slab" ) > $O/pytest_gpu_intime.log 2>&1
tail -4 $O/pytest_gpu_intime.log
for rep in 1 2 3; do
for mode in "" "--early-exchanges" "--early-exchanges --arrival-by-event"; do
  for r in 3; do
    timeout 300 python bench.py $mode --emulate-rank $r --of 8 --steps 30 --warmup 5 --sim-steps 8 > $O/emu_it.json 2> $O/emu_it.err || tail -3 $O/emu_it.err
    python -c "import json;d=json.load(open('$O/emu_it.json'));print('%-40s rank $r of 8: %.4f ms per solve, sim step %.1f us, %d exchanges' % ('${mode:-in time (sender tiles)}', d['ms_per_solve'], d['sim_step_us'] or 0, d['halo_exchanges_per_solve']))" | tee -a $O/intime_ab.txt
  done
done; done
for D in 0 10 25 50; do
  for mode in "" "--early-exchanges"; do
    timeout 300 python bench.py $mode --emulate-rank 3 --of 8 --wire-us $D --steps 20 --warmup 3 --sim-steps 0 > $O/emu_it.json 2> $O/emu_it.err || tail -3 $O/emu_it.err
    python -c "import json;d=json.load(open('$O/emu_it.json'));print('D=%3d us  %-24s %.4f ms per solve' % ($D, '${mode:-in time}', d['ms_per_solve']))" | tee -a $O/intime_wire.txt
  done
done
bash tools/r04/trace_emulate.sh intime --emulate-rank 3 --of 8 | sed -n 12,40p
