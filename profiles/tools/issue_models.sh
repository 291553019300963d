#!/bin/bash
# issue_models.sh <round> -- price the dominant kernel of every bench workload with the measured instruction-class table
# profiles/<round>/issue_costs.json (made on the GPU box by tools_dev/ubench/issue_cost) at the kernel's occupancy:
# profiles/<round>/<workload>_issue_model.json, read by bench.py's roofline.valu_issue.  No GPU needed (disassembles the built objects).
set -euo pipefail
ROUND=${1:-r03}
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/profiles/$ROUND/issue_costs.json
# workload : object : kernel regex : resident waves per SIMD (tools_dev/kernel_resources.sh + LDS per workgroup)
while IFS='|' read -r wl obj pat w; do
  python3 "$R/tools_dev/issue_model.py" "$C" "$obj" "$pat" "$w" "$R/profiles/$ROUND/${wl}_issue_model.json" > /dev/null
  python3 - "$R/profiles/$ROUND/${wl}_issue_model.json" <<'PY'
import json, sys
m = json.load(open(sys.argv[1]))
print("%-60s W=%d  %d static VALU  mean %.3f ns (plain %.3f)" % (m["kernel"][:60], m["waves_per_simd"], m["static_valu_instructions"], m["mean_ns_per_valu"], m["plain_v_add_f32"]["ns"]))
PY
done <<'LIST'
chain|aec|aec_near_kernel<2>|4
ns_aec_8k|aec|aec_near_kernel<1>|4
chain_8k|aec|aec_near_kernel<1>|4
ns|ns|ns_kernel<256, 1>|5
ns_agc_mix_32k|ns|ns_kernel<256, 2>|5
nsx|nsx|nsx_kernel<256, 1>|5
chain_fx|nsx|nsx_kernel<256, 1>|5
aecm|aecm|aecm_near_kernel|7
g711|g711|g711_encode_kernel<1>|8
mfft|mfft|mfft_regs_kernel<1, false, 9>|4
LIST
