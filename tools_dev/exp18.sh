#!/bin/bash
# the AEC core's two block counters with the stream: tests, A/B against the build before, cohort lines with --coalesce
set -euo pipefail
O=gpurun_out/exp18; mkdir -p $O
python -m pytest tests/test_aec_gpu.py tests/test_vs_reference_gpu.py tests/test_lifetime_gpu.py tests/test_cohorts_scale_gpu.py tests/test_coalesce_gpu.py tests/test_cadence_gpu.py tests/test_configs_gpu.py -q -m gpu -x 2>&1 | tail -3
for a in "" "--workload ns_aec_8k"; do for L in wmix_amd/libwmix_amd.so tools_dev/build/lib_prev2.so wmix_amd/libwmix_amd.so tools_dev/build/lib_prev2.so; do
  WMIX_AMD_LIB=$L python bench.py --no-cpu --steps 300 $a | python -c "import sys,json; d=json.load(sys.stdin); print(sys.argv[1], round(d[\"ms_per_step\"],4), round(d[\"roofline\"][\"avg_launch_ms\"],4), d[\"parity_checked\"][\"max_lsb\"])" "$L $a"
done; done
for a in "--cohorts 4096 --coalesce" "--cohorts 256 --coalesce" "--workload chain_8k --cohorts 256 --coalesce"; do
  n=$(echo $a | tr -d ' -'); python bench.py --no-cpu --steps 300 $a > $O/bench_$n.json
  python -c "import sys,json; d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), d['parity_checked']['max_lsb'], d['config'].get('coalesce'), d['config'].get('aec_host_control_plane_us_per_launch'))" $O/bench_$n.json "$a"
done
