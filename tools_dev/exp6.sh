#!/bin/bash
# round-4 session 6: cohort-sorted, XCD-aware stream order of the near kernel -- bench --cohorts with and without it, then the GPU suite
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp6; mkdir -p $O
for cfg in "1 arrival" "256 arrival" "256 interleaved" "4096 arrival" "4096 interleaved"; do
  set -- $cfg
  python bench.py --no-cpu --steps 300 --cohorts $1 --cohort-layout $2 > $O/bench_${1}_${2}.json 2> $O/err.txt || tail -5 $O/err.txt
  if [ $1 != 1 ]; then WMIX_AMD_AEC_NO_ORDER=1 python bench.py --no-cpu --steps 300 --cohorts $1 --cohort-layout $2 > $O/bench_${1}_${2}_noorder.json 2> $O/err.txt || tail -5 $O/err.txt; fi
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/exp6/bench_*.json")):
    try:
        d=json.load(open(f)); print(f, "%.4g"%d["value"], "%.4f"%d["ms_per_step"], "near %.4f"%d["roofline"]["avg_launch_ms"], d["parity_checked"]["max_lsb"], d["config"].get("aec_host_control_plane_us_per_launch"))
    except Exception as e: print(f, "ERR", e)
PY
python -m pytest tests -q -m gpu > $O/gpu_suite.txt 2>&1; tail -8 $O/gpu_suite.txt
