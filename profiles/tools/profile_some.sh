#!/bin/bash
# profile_some.sh <round> <workload:streams> ... -- profile_workload.sh for the named workloads only, e.g. after a kernel change:
#   gpurun -- 'bash profiles/tools/profile_some.sh r03 chain:65536 ns_aec_8k:131072'
ROUND=${1:-r03}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
for spec in "$@"; do
  wl=${spec%%:*}; n=${spec#*:}
  echo "== $wl"; bash $R/profiles/tools/profile_workload.sh $ROUND $wl $n > $R/gpurun_out/prof_$wl.log 2>&1; tail -1 $R/gpurun_out/prof_$wl.log
done
