#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp10; mkdir -p $O
python -m pytest tests/test_fft_gpu.py tests/test_aec_gpu.py tests/test_ns_gpu.py tests/test_vs_reference_gpu.py tests/test_extremes_gpu.py tests/test_cohorts_scale_gpu.py -q -m gpu > $O/tests.txt 2>&1; tail -4 $O/tests.txt
bash tools_dev/ab.sh "chain ns_aec_8k" tools_dev/build/lib_prev.so wmix_amd/libwmix_amd.so tools_dev/build/lib_prev.so wmix_amd/libwmix_amd.so 2>&1 | grep -E "^chain|^ns_aec" | tee $O/ab.txt
