#!/bin/bash
# r04: (1) what the thin share's launch is made of -- ablations of the shipped NS = 10 kernel on 8192 x 1024 (wrong results, timing only:
#      d1 no lane shifts, d2 no rhs ring in LDS, d3 no global loads, d4 all three, d5 every tile on the interior path)
#      (2) arrival by flag vs by event, interleaved repetitions
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
for rep in 1 2 3; do
for v in ns10 ns10_d1 ns10_d2 ns10_d3 ns10_d4 ns10_d5; do
  ./tools/sor_clock_probe_$v 8192 1024 40 0 > $O/abl_$v.txt 2>&1
  echo "$v: $(head -1 $O/abl_$v.txt | sed 's/.*events //')  span $(grep 'launch span' $O/abl_$v.txt | sed 's/.*: //')  $(grep 'lifetime, shader' $O/abl_$v.txt)  clock $(grep 'shader clock' $O/abl_$v.txt | sed 's/.*median //;s/ .*//')" | tee -a $O/thin_ablation.txt
done; done
for rep in 1 2 3 4; do
for arr in 1 0; do
    timeout 300 python bench.py $([ $arr = 0 ] && echo --arrival-by-event) --emulate-rank 3 --of 8 --steps 30 --warmup 5 --sim-steps 8 > $O/emu_arr.json 2> $O/emu_arr.err || tail -3 $O/emu_arr.err
    python -c "import json;d=json.load(open('$O/emu_arr.json'));print('arrival by flag $arr  rank 3 of 8: %.4f ms per solve, sim step %.1f us' % (d['ms_per_solve'], d['sim_step_us'] or 0))" | tee -a $O/arrival_ab3.txt
done; done
