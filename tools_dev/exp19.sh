#!/bin/bash
# where the fixed-point chain's many-cohort cost sits: kernel trace of chain_fx with 1 and 256 cohorts
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp19; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in 1 256; do
  rocprofv3 --kernel-trace --stats -d $O/t$c -o p --output-format csv -- python3 $R/bench.py --no-cpu --workload chain_fx --cohorts $c --steps 100 > $O/bench_$c.log 2>&1
  python3 - $O/t$c $c <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/p_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = collections.defaultdict(list)
for r in rows:
    import re
    m = re.search(r'(aecm_\w+|nsx_kernel|agc_pipe_kernel|vad_pipe_kernel)', r['Kernel_Name'])
    by[m.group(1) if m else 'other'].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in by.items():
    if 'aecm' in k or 'nsx' in k:
        tail = v[-100:]
        print(sys.argv[2], k, len(v), 'last100 avg us', round(sum(tail) / len(tail) / 1e3, 1), 'min', min(tail) / 1e3, 'max', max(tail) / 1e3)
PY
  rm -rf $O/t$c
done
