#!/bin/bash
# round-4 session 3: scalable cohorts -- AEC/lifetime tests on the new plan format, the at-scale test, bench --cohorts lines
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp3; mkdir -p $O
python -m pytest tests/test_aec_gpu.py tests/test_lifetime_gpu.py tests/test_cohorts_scale_gpu.py tests/test_cadence_gpu.py tests/test_host_chain_gpu.py -q -m gpu > $O/tests.txt 2>&1; tail -15 $O/tests.txt
for n in 1 16 256 4096; do
  python bench.py --no-cpu --steps 300 --cohorts $n > $O/bench_cohorts_$n.json 2> $O/bench_$n.err || tail -5 $O/bench_$n.err
done
python bench.py --no-cpu --steps 300 --cohorts 256 --cohort-layout interleaved > $O/bench_cohorts_256_interleaved.json 2> $O/bench_256i.err || tail -5 $O/bench_256i.err
python bench.py --no-cpu --steps 300 --packets-per-step 2 --interval-ms 20 > $O/bench_iv20.json 2> $O/bench_iv20.err || tail -5 $O/bench_iv20.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/exp3/bench_*.json")):
    try:
        d=json.load(open(f)); print(f, "%.4g"%d["value"], "%.4f"%d["ms_per_step"], "host %.4f"%d["host_wall_ms_per_step"], d["stage_ms"], d["parity_checked"]["max_lsb"], d["config"].get("aec_host_control_plane_us_per_launch"))
    except Exception as e: print(f, "ERR", e)
PY
