#!/bin/bash
# kernel_isa.sh <object-stem> <kernel-pattern> [out.s] -- disassemble one gfx950 kernel of wmix_amd/csrc/build/<stem>.o and
# print its size, SGPR spill traffic (v_readlane / v_writelane) and the 15 most frequent mnemonics.  Works without a GPU.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=${WMX_TOOL_OBJDIR:-$ROOT/wmix_amd/csrc/build}
T=$(mktemp -d); trap 'rm -rf "$T"' EXIT
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy --dump-section .hip_fatbin="$T/fat.bin" "$OBJ/$1.o"
$B/clang-offload-bundler --unbundle --type=o --input="$T/fat.bin" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$T/dev.co"
$B/llvm-objdump -d --demangle "$T/dev.co" | awk -v pat="$2" '/^[0-9a-f]+ <.*>:$/ {p = ($0 ~ pat)} p' > "${3:-$T/k.s}"
K=${3:-$T/k.s}
echo "instructions: $(grep -cE '^\s+[a-z]' "$K")   readlane: $(grep -c v_readlane "$K")   writelane: $(grep -c v_writelane "$K")"
awk '{print $1}' "$K" | grep -E '^[vs]_|^ds_|^global|^buffer|^scratch' | sort | uniq -c | sort -rn | head -15
