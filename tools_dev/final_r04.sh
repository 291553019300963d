#!/bin/bash
# round-4 wrap-up on the GPU box: the whole GPU suite, smoke(), the driver-style line, then the profiles of the workloads whose kernels changed
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/final; mkdir -p $O
python -m pytest tests -q -m gpu > $O/gpu_suite.txt 2>&1; tail -4 $O/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_style_bench_line.json 2> $O/driver_err.txt; python -c "
import json; d=json.load(open('$O/driver_style_bench_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['parity_checked']['max_lsb'], d['cpu_baseline']['value'] if d['cpu_baseline'] else None)"
bash profiles/tools/profile_some.sh r04 chain:65536 chain_8k:131072 ns_aec_8k:131072 ns:4096 ns_agc_mix_32k:32768
