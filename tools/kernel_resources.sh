#!/bin/bash
# Register / spill / scratch summary of every gfx950 kernel in the given objects (default: every object of the library).
# usage: bash tools/kernel_resources.sh [file.o ...]
set -u
B=/opt/rocm/lib/llvm/bin
files=${@:-esp32-fluid-simulation_amd/lib/*.o}
T=$(mktemp -d)
for f in $files; do
  $B/llvm-objcopy --dump-section .hip_fatbin=$T/fb.bin $f 2>/dev/null || continue
  $B/clang-offload-bundler --unbundle --type=o --input=$T/fb.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co 2>/dev/null || continue
  $B/llvm-readelf --notes $T/k.co | grep -E "\.name:|\.vgpr_count|vgpr_spill|private_segment_fixed|sgpr_spill|\.sgpr_count|group_segment_fixed" | paste - - - - - - - \
    | sed 's/ \+/ /g;s/\.private_segment_fixed_size/scratch/;s/\.group_segment_fixed_size/lds/;s/_ZN3sfl12_GLOBAL__N_1//' | awk -v f=$(basename $f) '{print f": "$0}' | cut -c1-330
done
rm -rf $T
