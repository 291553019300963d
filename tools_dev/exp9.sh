#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp9; mkdir -p $O
python bench.py --no-cpu --steps 300 --workload chain_8k > $O/chain_8k.json 2> $O/err.txt || tail -3 $O/err.txt
python bench.py --no-cpu --steps 300 --workload chain_8k --interval-ms 20 --packets-per-step 2 > $O/chain_8k_iv20.json 2> $O/err.txt || tail -3 $O/err.txt
python bench.py --no-cpu --steps 300 --cohorts 256 --cohort-layout interleaved > $O/chain_cohorts_256_interleaved.json 2> $O/err.txt || tail -3 $O/err.txt
python bench.py --no-cpu --steps 300 --cohorts 4096 --cohort-layout interleaved > $O/chain_cohorts_4096_interleaved.json 2> $O/err.txt || tail -3 $O/err.txt
python bench.py --no-cpu --steps 300 --cohorts 16 > $O/chain_cohorts_16_arrival.json 2> $O/err.txt || tail -3 $O/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/exp9/*.json")):
    try:
        d=json.load(open(f)); print(f, "%.4g"%d["value"], "%.4f"%d["ms_per_step"], {k:round(v,4) for k,v in d["stage_ms"].items()}, d["parity_checked"]["max_lsb"])
    except Exception as e: print(f, "ERR", e)
PY
