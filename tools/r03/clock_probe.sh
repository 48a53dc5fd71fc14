#!/bin/bash
# r03: clock of sor_fused_kernel by three independent methods (runs on the GPU box via gpurun)
set -u
export TMPDIR=/tmp
O=gpurun_out/r03_clock
mkdir -p $O
./tools/sor_clock_probe_ns16 8192 8192 40 0 $O/ns16_8192.csv > $O/ns16_8192.txt 2>&1
cat $O/ns16_8192.txt
./tools/sor_clock_probe_ns10 8192 1024 40 0 $O/ns10_slab1024.csv > $O/ns10_slab1024.txt 2>&1
cat $O/ns10_slab1024.txt
./tools/sor_clock_probe_ns16 8192 1024 40 0 $O/ns16_slab1024.csv > $O/ns16_slab1024.txt 2>&1
cat $O/ns16_slab1024.txt
# sclk as the driver reports it while the kernel loops for 4 s
./tools/sor_clock_probe_ns16 8192 8192 10 0 "" 4 > $O/loop.txt 2>&1 &
LP=$!
sleep 1.5
for k in 1 2 3 4; do
  (rocm-smi --showclocks 2>&1 | grep -iE "sclk|mclk|fclk" | head -6; rocm-smi --showpower 2>&1 | grep -i power | head -3) >> $O/smi.txt 2>&1
  sleep 0.4
done
wait $LP
cat $O/loop.txt
cat $O/smi.txt | head -40
amd-smi metric --clock --power 2>&1 | head -60 > $O/amdsmi_idle.txt
