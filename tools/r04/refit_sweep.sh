#!/bin/bash
# r04: fuse depth x halo depth x rows per tile of the emulated rank 3 of 8 (BASELINE config 4) with the device-side
# arrival count in place; ms per solve | exchanges per solve
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
out=$O/refit_sweep.txt
for fuse in 8 10 12; do
  for halo in 0 24 32 40 48 64; do
    [ $halo -ne 0 ] && [ $halo -lt $((2 * fuse)) ] && continue
    for rows in 0; do
      timeout 200 python bench.py --emulate-rank 3 --of 8 --fuse $fuse --sor-halo $halo --sor-rows $rows --steps 20 --warmup 3 --sim-steps 0 > $O/refit.json 2> $O/refit.err || { tail -2 $O/refit.err; continue; }
      python -c "import json;d=json.load(open('$O/refit.json'));print('fuse %2d  halo %2d  rows %2d : %.4f ms per solve  %d launches  %d exchanges' % ($fuse, $halo, $rows, d['ms_per_solve'], d['sor_launches_per_solve'], d['halo_exchanges_per_solve']))" | tee -a $out
    done
  done
done
for rows in 24 32 36 41 48 56; do
  timeout 200 python bench.py --emulate-rank 3 --of 8 --sor-rows $rows --steps 20 --warmup 3 --sim-steps 0 > $O/refit.json 2> $O/refit.err || { tail -2 $O/refit.err; continue; }
  python -c "import json;d=json.load(open('$O/refit.json'));print('fuse auto halo auto rows %2d : %.4f ms per solve  %d launches  %d exchanges' % ($rows, d['ms_per_solve'], d['sor_launches_per_solve'], d['halo_exchanges_per_solve']))" | tee -a $out
done
