#!/bin/bash
# round-4 session 2: the daemon's cadence (interval 20) -- new tests, bench lines at interval 10 / 20, legacy adapter latency
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp2; mkdir -p $O
python -m pytest tests/test_cadence_gpu.py -q -m gpu -x > $O/cadence_tests.txt 2>&1; tail -15 $O/cadence_tests.txt
python tools_dev/legacy_latency.py > $O/legacy_latency.jsonl 2> $O/legacy_latency.err; cat $O/legacy_latency.jsonl; tail -3 $O/legacy_latency.err
for args in "--packets-per-step 1" "--packets-per-step 2" "--packets-per-step 2 --interval-ms 20"; do
  python bench.py --no-cpu --steps 300 $args > $O/bench_$(echo $args | tr -d ' -').json 2> $O/bench.err || tail -5 $O/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/exp2/bench_*.json")):
    d=json.load(open(f)); print(f, d["value"], d["ms_per_step"], d["stage_ms"], d["parity_checked"]["max_lsb"])
PY
