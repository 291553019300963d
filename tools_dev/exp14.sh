#!/bin/bash
# coalescing: whole GPU suite, soak under churn, bench lines with --cohorts N --coalesce
set -euo pipefail
O=gpurun_out/exp14; mkdir -p $O
python -m pytest tests -q -m gpu -x 2>&1 | tail -4 | tee $O/gpu_suite.txt
python tools_dev/churn_soak.py --streams 4096 --ticks 3000 --coalesce | tee $O/soak_coalesce.json
python tools_dev/churn_soak.py --streams 2048 --ticks 2500 --freq 8000 --coalesce --seed 3 | tee $O/soak_coalesce_8k.json
for a in "--cohorts 1" "--cohorts 256" "--cohorts 256 --coalesce" "--cohorts 4096 --cohort-layout interleaved" "--cohorts 4096 --cohort-layout interleaved --coalesce" "--cohorts 4096 --coalesce"; do
  n=$(echo $a | tr -d ' -'); python bench.py --no-cpu --steps 300 $a > $O/bench_$n.json
  python -c "import sys,json; d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), d['parity_checked']['max_lsb'], d['config'].get('coalesce'), d['config'].get('aec_host_control_plane_us_per_launch'))" $O/bench_$n.json "$a"
done
