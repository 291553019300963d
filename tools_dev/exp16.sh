#!/bin/bash
set -euo pipefail
O=gpurun_out/exp16; mkdir -p $O
python -m pytest tests/test_cohorts_scale_gpu.py tests/test_coalesce_gpu.py tests/test_lifetime_gpu.py -q -m gpu -x 2>&1 | tail -3
for a in "--cohorts 1" "--cohorts 16" "--cohorts 256" "--cohorts 256 --cohort-layout interleaved" "--cohorts 256 --coalesce" "--cohorts 4096" "--cohorts 4096 --cohort-layout interleaved" "--cohorts 4096 --cohort-layout interleaved --coalesce" "--cohorts 4096 --coalesce" "--workload chain_8k --cohorts 256" "--workload chain_8k --cohorts 256 --coalesce"; do
  n=$(echo $a | tr -d ' -'); python bench.py --no-cpu --steps 300 $a > $O/bench_$n.json
  python -c "import sys,json; d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), d['parity_checked']['max_lsb'], d['config'].get('coalesce'), d['config'].get('aec_host_control_plane_us_per_launch'))" $O/bench_$n.json "$a"
done
