#!/bin/bash
# r03: boundary-strip tiles spread over the XCDs (vs all behind the inner tiles), boundary row cost 9 / 10 / 11 sixteenths
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_spread
for rep in 1 2 3 4; do
  l="rep $rep 8192^2:"
  for v in _nospread "" _cost9 _cost11; do
    a=$(./tools/sor_clock_probe_ns16$v 8192 8192 20 0 | grep -E "waves traced" | sed 's/.*events //;s/ us//')
    l="$l ${v:-spread} $a"
  done
  b0=$(./tools/sor_clock_probe_ns10_nospread 8192 1024 60 0 | grep -E "waves traced" | sed 's/.*events //;s/ us//')
  b1=$(./tools/sor_clock_probe_ns10 8192 1024 60 0 | grep -E "waves traced" | sed 's/.*events //;s/ us//')
  c0=$(./tools/sor_clock_probe_ns16_nospread 8192 2048 40 0 | grep -E "waves traced" | sed 's/.*events //;s/ us//')
  c1=$(./tools/sor_clock_probe_ns16 8192 2048 40 0 | grep -E "waves traced" | sed 's/.*events //;s/ us//')
  d0=$(./tools/sor_clock_probe_ns16_nospread 16384 16384 8 0 | grep -E "waves traced" | sed 's/.*events //;s/ us//')
  d1=$(./tools/sor_clock_probe_ns16 16384 16384 8 0 | grep -E "waves traced" | sed 's/.*events //;s/ us//')
  echo "$l | 8192x1024 NS10: $b0 -> $b1 | 8192x2048 NS16: $c0 -> $c1 | 16384^2: $d0 -> $d1" | tee -a gpurun_out/r03_spread/ab.txt
done
./tools/sor_clock_probe_ns16 8192 8192 30 0 | grep -E "last wave|wave end" | tee -a gpurun_out/r03_spread/ab.txt
