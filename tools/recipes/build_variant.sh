#!/bin/bash
# Compile-time VARIANTS for A/B measurements (never the product): hipcc cross-compiles here, the results travel to the GPU box
# with the snapshot under build/ (git-ignored).
#   lib   <name> "<extra hipcc flags>" [sources, default "sor_fused.hip:2,5"]
#         -> build/variants/<name>/libsfl_hip.so: the product's objects with the named sources rebuilt with the flags
#            (sor_fused.hip:<groups> rebuilds those fuse-depth groups, 2 = NS 10, 5 = NS 16; any other csrc file by name);
#            run with  python tools/with_lib.py build/variants/<name>/libsfl_hip.so bench.py ...  or LIB=<path> in bench_sweep.sh
#   probe <name> <NS> "<extra flags>"
#         -> build/probes/sor_clock_probe_<name>: tools/sor_clock_probe.hip (the fused kernel with per-wave clocks; the
#            SFL_PROBE_* ablations and timing mocks are legal only here) at fuse depth NS
# e.g.  bash tools/recipes/build_variant.sh probe ns10_vpipe 10 "-DSFL_PROBE_COOP=3"
#       bash tools/recipes/build_variant.sh lib seam_early "-DSEAM_DYE_LOADS=1" advect_tiled.hip
set -eu
cd "$(dirname "$0")/../.."
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -fvisibility=hidden -Wno-unused-parameter"
SRC=esp32-fluid-simulation_amd/csrc; LIB=esp32-fluid-simulation_amd/lib
mode=$1; name=$2
if [ $mode = probe ]; then
  ns=$3; flags=${4:-}
  case $ns in 2|4|6) g=0;; 8) g=1;; 10) g=2;; 12) g=3;; 14) g=4;; 16) g=5;; *) echo "fuse depth $ns?"; exit 1;; esac
  mkdir -p build/probes
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -DSFL_NS_GROUP=$g -DPROBE_NS=$ns $flags tools/sor_clock_probe.hip -o build/probes/sor_clock_probe_$name
  echo built build/probes/sor_clock_probe_$name; exit 0
fi
flags=$3; sources=${4:-"sor_fused.hip:2,5"}
OUT=build/variants/$name; mkdir -p $OUT
replaced=""
for s in $sources; do
  file=${s%%:*}
  if [ $file = sor_fused.hip ]; then
    groups=${s#*:}; [ "$groups" = "$s" ] && groups="2,5"
    for g in ${groups//,/ }; do for p in 0 1; do for f in 0 1; do
      hipcc $F $flags -DSFL_NS_GROUP=$g -DSFL_DX_PART=$p -DSFL_FOLD_PART=$f -c $SRC/sor_fused.hip -o $OUT/sor_fused_g${g}_p${p}_f$f.o &
      replaced="$replaced sor_fused_g${g}_p${p}_f$f.o"
    done; done; done
  else
    b=$(basename ${file%.*}).o
    case $file in *.cpp) x="-x hip";; *) x="";; esac
    hipcc $F $flags $x -c $SRC/$file -o $OUT/$b &
    replaced="$replaced $b"
  fi
done
wait
objs=""
for o in $LIB/*.o; do b=$(basename $o); case " $replaced " in *" $b "*) objs="$objs $OUT/$b";; *) objs="$objs $o";; esac; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libsfl_hip.so $objs -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built $OUT/libsfl_hip.so "(rebuilt:$replaced)"
