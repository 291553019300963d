#!/bin/bash
# noise rows by per-stream block count: AEC + cohort + coalesce tests, then A/B vs HEAD~ and the cohort lines
set -euo pipefail
O=gpurun_out/exp15; mkdir -p $O
python -m pytest tests/test_cohorts_scale_gpu.py tests/test_coalesce_gpu.py tests/test_lifetime_gpu.py -q -m gpu -x 2>&1 | tail -3
for a in "" "--workload ns_aec_8k"; do for L in wmix_amd/libwmix_amd.so tools_dev/build/lib_head.so wmix_amd/libwmix_amd.so tools_dev/build/lib_head.so; do
  WMIX_AMD_LIB=$L python bench.py --no-cpu --steps 300 $a | python -c "import sys,json; d=json.load(sys.stdin); print(sys.argv[1], round(d[\"ms_per_step\"],4), round(d[\"roofline\"][\"avg_launch_ms\"],4), d[\"parity_checked\"][\"max_lsb\"])" "$L $a"
done; done
for a in "--cohorts 256" "--cohorts 256 --coalesce" "--cohorts 4096 --cohort-layout interleaved" "--cohorts 4096 --cohort-layout interleaved --coalesce" "--cohorts 4096 --coalesce"; do
  n=$(echo $a | tr -d ' -'); python bench.py --no-cpu --steps 300 $a > $O/bench_$n.json
  python -c "import sys,json; d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), d['parity_checked']['max_lsb'], d['config'].get('coalesce'), d['config'].get('aec_host_control_plane_us_per_launch'))" $O/bench_$n.json "$a"
done
