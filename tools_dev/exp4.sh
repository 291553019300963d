#!/bin/bash
# round-4 session 4: tolerance builds of the near kernel -- steady-state LSB (tol_check) and A/B timing
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp4; mkdir -p $O
B=tools_dev/build
for L in $B/lib_fma.so $B/lib_fma_div.so; do WMIX_AMD_LIB=$R/$L python tools_dev/tol_check.py 64 >> $O/tol_check.jsonl 2>> $O/tol.err; done
cat $O/tol_check.jsonl
bash tools_dev/ab.sh chain wmix_amd/libwmix_amd.so $B/lib_fma.so $B/lib_fma_div.so wmix_amd/libwmix_amd.so > $O/ab.txt 2>&1; cat $O/ab.txt
