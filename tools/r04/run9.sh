#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
for rep in 1 2 3; do
for v in "ns16 8192 8192" "ns16_coop2 8192 8192" "ns16 8192 2048" "ns16_coop2 8192 2048"; do
  set -- $v
  ./tools/sor_clock_probe_$1 $2 $3 40 0 > $O/coop_$1_$3.txt 2>&1
  echo "$1 $2x$3: $(head -1 $O/coop_$1_$3.txt | sed 's/.*events //')  span $(grep 'launch span' $O/coop_$1_$3.txt | sed 's/.*: //')  $(grep 'lifetime, shader' $O/coop_$1_$3.txt)  clock $(grep 'shader clock' $O/coop_$1_$3.txt | sed 's/.*median //;s/ .*//')" | tee -a $O/coop_mock.txt
done; done
