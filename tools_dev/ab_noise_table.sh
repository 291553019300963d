#!/bin/bash
# A/B of the AEC comfort-noise phases: table rows indexed by the stream's block count (round 4) against the near kernel
# computing them itself (jump-ahead on the scalar unit + one gather from the 32 768-entry cosf / sinf table).
# Run on the GPU box from the repository root; writes gpurun_out/r05/ab_noise_*.
set -e
O=gpurun_out/r05
mkdir -p $O
python -m pytest tests/test_aec_gpu.py tests/test_lifetime_gpu.py -x -q -m gpu > $O/ab_noise_tests_table.log 2>&1
WMIX_AMD_AEC_NO_NOISE_TABLE=1 python -m pytest tests/test_aec_gpu.py tests/test_coalesce_gpu.py tests/test_lifetime_gpu.py -x -q -m gpu > $O/ab_noise_tests_compute.log 2>&1
for rep in 1 2; do
  python3 bench.py --steps 400 --no-cpu --no-configs > $O/ab_noise_chain_table_$rep.json 2>$O/err.log
  WMIX_AMD_AEC_NO_NOISE_TABLE=1 python3 bench.py --steps 400 --no-cpu --no-configs > $O/ab_noise_chain_compute_$rep.json 2>$O/err.log
done
python3 bench.py --workload ns_aec_8k --steps 400 --no-cpu > $O/ab_noise_8k_table.json 2>$O/err.log
WMIX_AMD_AEC_NO_NOISE_TABLE=1 python3 bench.py --workload ns_aec_8k --steps 400 --no-cpu > $O/ab_noise_8k_compute.json 2>$O/err.log
tail -2 $O/ab_noise_tests_table.log $O/ab_noise_tests_compute.log
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05/ab_noise_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "ms_per_step %.4f" % d["ms_per_step"], "near %.4f" % d["roofline"]["avg_launch_ms"], "agc", d["stage_ms"].get("agc"), "parity", d["parity_checked"]["max_lsb"])
PY
