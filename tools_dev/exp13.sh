#!/bin/bash
# per-stream comfort-noise generator: AEC tests + A/B of the chain against HEAD
set -euo pipefail
mkdir -p gpurun_out/exp13
python -m pytest tests/test_aec_gpu.py tests/test_vs_reference_gpu.py tests/test_lifetime_gpu.py tests/test_cohorts_scale_gpu.py -q -m gpu -x 2>&1 | tail -3
for a in "" "--cohorts 256" "--workload ns_aec_8k"; do for L in wmix_amd/libwmix_amd.so tools_dev/build/lib_head.so wmix_amd/libwmix_amd.so tools_dev/build/lib_head.so; do
  WMIX_AMD_LIB=$L python bench.py --no-cpu --steps 300 $a | python -c "import sys,json; d=json.load(sys.stdin); print(sys.argv[1], round(d[\"ms_per_step\"],4), round(d[\"roofline\"][\"avg_launch_ms\"],4), d[\"parity_checked\"][\"max_lsb\"])" "$L $a"
done; done
