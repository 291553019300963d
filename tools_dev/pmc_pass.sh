#!/bin/bash
# developer helper: one extra rocprofv3 counter pass over the default bench command, per-kernel means printed
#   gpurun -- 'bash tools_dev/pmc_pass.sh icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH'
# WMX_PMC_WORKLOAD=<name> selects another bench workload.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" -d "$OUT" -o p --output-format csv -- python3 $R/bench.py --no-cpu --steps 20 --warmup 4 ${WMX_PMC_WORKLOAD:+--workload $WMX_PMC_WORKLOAD} > "$OUT/log.txt" 2>&1
cd "$R"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        m = re.search(r"(\w+_kernel(<[^>]*>)?)", r["Kernel_Name"]); k = m.group(1) if m else "other"
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if "_kernel" in k:
        print(k, {n: round(sum(v[-20:]) / len(v[-20:])) for n, v in c.items()})
PY
