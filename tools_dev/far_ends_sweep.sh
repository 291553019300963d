#!/bin/bash
# The non-shared far-end regime (round-4 VERDICT "next" 4): the chain at 65 536 streams with N distinct far-ends, N = 1 .. 65 536.
# Run on the GPU box from the repository root; one JSON line per N under gpurun_out/r06/.
O=gpurun_out/r06
mkdir -p $O
TAG=${1:-a}
for n in 1 256 4096 65536; do
  timeout -k 10 300 python3 bench.py --far-ends $n --steps 200 --no-cpu --no-configs > $O/far_ends_${TAG}_$n.json 2> $O/far_ends_${TAG}_$n.err || { echo "N=$n failed"; tail -n 5 $O/far_ends_${TAG}_$n.err; break; }
done
python3 - <<PY
import json,glob
for n in (1,256,4096,65536):
    try:
        d=json.loads(open("$O/far_ends_${TAG}_%d.json" % n).read().strip().splitlines()[-1])
    except Exception as e:
        print(n, "no line", e); continue
    st=d["stage_ms"]
    print("N=%6d step %.4f ms host_wall %.4f near %.4f far %.4f frac %.4f ctl_us %s parity %s" % (n, d["ms_per_step"], d["host_wall_ms_per_step"], st.get("aec_near_kernel (timed region)",0), st.get("aec_far_kernel (timed region)",0), d["roofline"]["frac"], d["config"]["aec_host_control_plane_us_per_launch"], d["parity_checked"]["max_lsb"]))
PY
