#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "overlapped or arrival or virtual or config4 or emulated or slab" ) 2>&1 | tail -2
for rep in 1 2 3; do
for v in product nostrip; do
    if [ $v = product ]; then run="python bench.py"; else run="python tools/with_lib.py build/variants/$v/libsfl_hip.so bench.py"; fi
    timeout 300 $run --emulate-rank 3 --of 8 --steps 30 --warmup 5 --sim-steps 8 > $O/emu_it.json 2> $O/emu_it.err || tail -3 $O/emu_it.err
    python -c "import json;d=json.load(open('$O/emu_it.json'));print('%-10s rank 3 of 8: %.4f ms per solve, sim step %.1f us' % ('$v', d['ms_per_solve'], d['sim_step_us'] or 0))" | tee -a $O/stripmajor_ab.txt
done; done
bash tools/r04/trace_emulate.sh stripmajor --emulate-rank 3 --of 8 | sed -n 14,30p
