#!/bin/bash
# Builds a VARIANT of the product library (compile-time probes / experiments) under build/variants/<name>/libsfl_hip.so.
# Only the fused-SOR objects the experiment needs are rebuilt with the extra flags (default: fuse groups 2 and 5 =
# NS 10 and 16, the two automatic depths); everything else is linked from the product's own objects.
# usage: bash tools/r04/build_variant.sh <name> "<extra hipcc flags>" [groups, default "2 5"]
set -eu
cd "$(dirname "$0")/../.."
name=$1; flags=$2; groups=${3:-"2 5"}
SRC=esp32-fluid-simulation_amd/csrc; LIB=esp32-fluid-simulation_amd/lib; OUT=build/variants/$name
mkdir -p $OUT
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -fvisibility=hidden -Wno-unused-parameter"
objs=""
for o in $LIB/*.o; do
  b=$(basename $o); use=$o
  for g in $groups; do
    for p in 0 1; do
      if [ $b = sor_fused_g${g}_p$p.o ]; then
        use=$OUT/$b
        hipcc $F $flags -DSFL_NS_GROUP=$g -DSFL_DX_PART=$p -c $SRC/sor_fused.hip -o $use &
      fi
    done
  done
  objs="$objs $use"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libsfl_hip.so $objs -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built $OUT/libsfl_hip.so
