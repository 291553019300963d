#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp8; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lay in arrival interleaved; do
  rocprofv3 --kernel-trace --stats -d $O/$lay -o p --output-format csv -- python3 $R/bench.py --no-cpu --steps 100 --cohorts 256 --cohort-layout $lay > $O/bench_$lay.json 2> $O/err_$lay.txt
  python3 - $O/$lay <<'PY'
import csv,glob,sys,collections
p=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(p)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last 100+16 steps region: take the final 2000 kernel rows, group by name
acc=collections.defaultdict(list)
sel=rows[-(116*6):-(16*6)]
for r in sel:
    n=r["Kernel_Name"]; k=n.split("(")[0][-40:]
    acc[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in acc.items(): print(k, len(v), "avg %.1f us"%(sum(v)/len(v)))
t0=int(sel[0]["Start_Timestamp"]); t1=int(sel[-1]["End_Timestamp"]); print("span per step %.1f us"%((t1-t0)/1e3/100))
# gaps: time between consecutive kernels on the timeline (any stream)
busy=0; last=t0
for r in sel:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    if e>last: busy+=e-max(s,last); last=e
print("busy per step %.1f us"%(busy/1e3/100))
PY
done
