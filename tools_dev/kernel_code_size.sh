#!/bin/bash
# kernel_code_size.sh [pattern] -- bytes of machine code of every gfx950 kernel in wmix_amd/csrc/build/*.o (the instruction
# cache is 64 KB per two CUs: a hot loop larger than that refetches from L2 every iteration).  Works without a GPU.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=${WMX_TOOL_OBJDIR:-$ROOT/wmix_amd/csrc/build}
T=$(mktemp -d); trap 'rm -rf "$T"' EXIT
B=/opt/rocm/lib/llvm/bin
for o in "$OBJ"/*.o; do
  $B/llvm-objcopy --dump-section .hip_fatbin="$T/fat.bin" "$o" 2>/dev/null || continue
  $B/clang-offload-bundler --unbundle --type=o --input="$T/fat.bin" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$T/dev.co" 2>/dev/null || continue
  $B/llvm-readelf -s --wide "$T/dev.co" | awk '$4 == "FUNC" {printf "%8d %s\n", $3, $8}'
done | c++filt | grep -E "${1:-.}" | sort -n
