#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp7; mkdir -p $O
B=tools_dev/build
for L in $B/lib_fma_div.so $B/lib_fma.so; do WMIX_AMD_LIB=$R/$L python tools_dev/tol_check.py 256 3 >> $O/tol_check_ns_aec.jsonl 2>> $O/tol.err; done
cat $O/tol_check_ns_aec.jsonl
