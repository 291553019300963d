#!/bin/bash
# every bench workload through profile_workload.sh: gpurun -- 'bash profiles/tools/profile_all.sh r03'
ROUND=${1:-r05}; SPECS=${2:-"chain:65536 chain_8k:131072 chain_fx:65536 ns_aec_8k:131072 ns_agc_mix_32k:32768 ns:4096 g711:1048576 mfft:65536 nsx:65536 aecm:65536 rtp_chain:131072 conference:65536"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
for spec in $SPECS; do
  wl=${spec%%:*}; n=${spec#*:}
  echo "== $wl"; bash $R/profiles/tools/profile_workload.sh $ROUND $wl $n > $R/gpurun_out/prof_$wl.log 2>&1; tail -1 $R/gpurun_out/prof_$wl.log
done
# the instruction-class price table behind roofline.valu_issue (then, on any machine: bash profiles/tools/issue_models.sh $ROUND)
mkdir -p $R/gpurun_out/$ROUND && (cd $R/tools_dev/ubench && ./issue_cost 1 2 4 5 7 8 > $R/gpurun_out/$ROUND/issue_costs.json 2> $R/gpurun_out/$ROUND/issue_costs.err)
