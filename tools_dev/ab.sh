#!/bin/bash
# ab.sh "<workloads>" <lib> [<lib> ...] -- bench.py --no-cpu for each workload against each library build, on the same box:
# prints ms/step, the dominant kernel's launch ms and the parity result (max LSB difference against the oracle).
set -euo pipefail
WL=$1; shift
for w in $WL; do for L in "$@"; do
  echo "$w $L $(WMX_TOOL_LIB=$L python tools_dev/bench_lib.py --workload $w --no-cpu --steps 300 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["parity_checked"]["max_lsb"])')"
done; done
