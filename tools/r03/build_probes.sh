#!/bin/bash
# Builds the variants of tools/sor_clock_probe.hip that the tools/r03/*.sh experiment scripts run (hipcc cross-compiles here, the
# binaries travel to the GPU box with the snapshot; they are git-ignored).  Usage: bash tools/r03/build_probes.sh [all|base|diag|prio]
set -u
cd "$(dirname "$0")/../.."
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math"
what=${1:-base}
b() { out=$1; shift; hipcc $F "$@" tools/sor_clock_probe.hip -o tools/$out 2>&1 | grep -E "error" & }
grp() { case $1 in 8) echo 1;; 10) echo 2;; 12) echo 3;; 14) echo 4;; 16) echo 5;; *) echo 0;; esac; }
if [ $what = base ] || [ $what = all ]; then   # the shipped kernel at every compiled depth the scripts use
  for ns in 8 10 12 14 16; do b sor_clock_probe_ns$ns -DSFL_NS_GROUP=$(grp $ns) -DPROBE_NS=$ns; done; wait
fi
if [ $what = diag ] || [ $what = all ]; then   # tools/r03/diag_probe.sh: ablations, all with the rotation forced on
  P="-DSFL_NS_GROUP=5 -DPROBE_NS=16 -DSFL_PRIO_FORCE=1"
  b sor_clock_probe_ns16_prio4 $P
  b sor_clock_probe_ns16_d1 $P -DSFL_PROBE_SHIFT=1
  b sor_clock_probe_ns16_d2 $P -DSFL_PROBE_NO_LDS=1
  b sor_clock_probe_ns16_d3 $P -DSFL_PROBE_NO_LOAD=1
  b sor_clock_probe_ns16_d4 $P -DSFL_PROBE_SHIFT=1 -DSFL_PROBE_NO_LDS=1 -DSFL_PROBE_NO_LOAD=1
  b sor_clock_probe_ns16_d5 $P -DSFL_PROBE_NO_EDGE=1
  wait
fi
if [ $what = prio ] || [ $what = all ]; then   # tools/r03/prio_sweep.sh, prio_sizes.sh: levels x rows per level, rotation off
  for cfg in "1 6" "4 6" "4 2" "4 1" "3 6" "3 2" "3 1" "2 3"; do set -- $cfg
    b sor_clock_probe_ns16_L$1R$2 -DSFL_NS_GROUP=5 -DPROBE_NS=16 -DSFL_PRIO_LEVELS=$1 -DSFL_PRIO_ROWS=$2 -DSFL_PRIO_FORCE=$([ $1 = 1 ] && echo 0 || echo 1)
    b sor_clock_probe_ns10_L$1R$2 -DSFL_NS_GROUP=2 -DPROBE_NS=10 -DSFL_PRIO_LEVELS=$1 -DSFL_PRIO_ROWS=$2 -DSFL_PRIO_FORCE=$([ $1 = 1 ] && echo 0 || echo 1)
  done; wait
  b sor_clock_probe_ns16_noprio -DSFL_NS_GROUP=5 -DPROBE_NS=16 -DSFL_PRIO_FORCE=0
  b sor_clock_probe_ns16_nont -DSFL_NS_GROUP=5 -DPROBE_NS=16 "-DSFL_NT_STORE_CELLS=(1ull<<40)"
  wait
fi
ls tools | grep sor_clock_probe_ns
