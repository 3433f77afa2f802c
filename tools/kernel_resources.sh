#!/bin/bash
# tools/kernel_resources.sh <object.o> -- registers / spills / LDS of every gfx950 kernel in a hipcc object (the code-object metadata notes)
set -e
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$1" $T/copy.o   # (an explicit output: without one llvm-objcopy rewrites its input in place and make relinks)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.co | awk '
  function flush() { if (name != "") printf "%-110s vgpr %3s sgpr %3s vspill %3s sspill %3s scratch %4s lds %6s\n", name, v, s, vs, ss, p, g; name = "" }
  /^ *- \.agpr_count:/ {flush()}      # (the keys of a kernel are sorted: .agpr_count opens its entry, .name comes later)
  /\.name:/ {name=$2}
  /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.vgpr_spill_count:/ {vs=$2} /\.sgpr_spill_count:/ {ss=$2}
  /\.private_segment_fixed_size:/ {p=$2} /\.group_segment_fixed_size:/ {g=$2}
  END {flush()}'
[ -n "$2" ] && cp $T/k.co "$2"
rm -rf $T
