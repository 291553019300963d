#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp11; mkdir -p $O
python bench.py --no-cpu --steps 300 --interval-ms 20 --packets-per-step 2 > $O/chain_iv20.json 2> $O/err.txt || tail -3 $O/err.txt
python bench.py --no-cpu --steps 300 --workload chain_8k --interval-ms 20 --packets-per-step 2 > $O/chain_8k_iv20.json 2> $O/err.txt || tail -3 $O/err.txt
python bench.py --no-cpu --steps 300 --workload chain_fx --interval-ms 20 --packets-per-step 2 > $O/chain_fx_iv20.json 2> $O/err.txt || tail -3 $O/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/exp11/*.json")):
    d=json.load(open(f)); print(f, "%.4g"%d["value"], "%.4f"%d["ms_per_step"], {k:round(v,4) for k,v in d["stage_ms"].items()}, d["parity_checked"]["max_lsb"])
PY
bash profiles/tools/profile_some.sh r04 ns_agc_mix_32k:32768
