#!/bin/bash
# A/B of a near-kernel change against tools_dev/build/lib_prev.so (the committed build): AEC parity tests, then the chain and the 8 kHz pair twice each
set -uo pipefail
python -m pytest tests/test_aec_gpu.py tests/test_fft_gpu.py tests/test_vs_reference_gpu.py -q -m gpu 2>&1 | tail -2
bash tools_dev/ab.sh "chain ns_aec_8k" tools_dev/build/lib_prev.so wmix_amd/libwmix_amd.so tools_dev/build/lib_prev.so wmix_amd/libwmix_amd.so 2>&1 | grep -E "^chain|^ns_aec"
