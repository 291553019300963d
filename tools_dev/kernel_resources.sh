#!/bin/bash
# kernel_resources.sh [pattern] -- VGPR / SGPR / LDS / scratch of every gfx950 kernel in wmix_amd/csrc/build/*.o whose
# (demangled) name matches `pattern` (default: all), read from the code objects' metadata.  Works without a GPU.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=${WMX_TOOL_OBJDIR:-$ROOT/wmix_amd/csrc/build}
T=$(mktemp -d); trap 'rm -rf "$T"' EXIT
B=/opt/rocm/lib/llvm/bin
for o in "$OBJ"/*.o; do
  $B/llvm-objcopy --dump-section .hip_fatbin="$T/fat.bin" "$o" 2>/dev/null || continue
  $B/clang-offload-bundler --unbundle --type=o --input="$T/fat.bin" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$T/dev.co" 2>/dev/null || continue
  $B/llvm-readelf --notes "$T/dev.co" 2>/dev/null | awk '
    /\.group_segment_fixed_size:/ {lds=$2} /\.name:/ {name=$2} /\.private_segment_fixed_size:/ {scr=$2} /\.sgpr_count:/ {sg=$2}
    /\.vgpr_count:/ {printf "%s vgpr %3d sgpr %3d lds %6d scratch %d\n", name, $2, sg, lds, scr}'
done | c++filt | grep -E "${1:-.}" | sort
