#!/bin/bash
# round-4 session 5: helper-wave hand-off cost (timing-only builds) + the whole GPU suite on the current tree
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp5; mkdir -p $O
B=tools_dev/build
bash tools_dev/ab.sh chain wmix_amd/libwmix_amd.so $B/lib_exp2.so $B/lib_exp2bar.so $B/lib_bar.so wmix_amd/libwmix_amd.so 2>&1 | grep chain > $O/ab_helper.txt; cat $O/ab_helper.txt
python -m pytest tests -q -m gpu -x > $O/gpu_suite.txt 2>&1; tail -8 $O/gpu_suite.txt
