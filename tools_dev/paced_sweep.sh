#!/bin/bash
# paced_sweep.sh <outdir> -- on the GPU box: what bounds a paced tick (round 6).  Back-to-back reference at the sub-batch sizes, then the
# paced tick resident and from host memory over sub-batch size x compute streams.
set -uo pipefail
out=${1:?outdir}; mkdir -p $out
R=$(cd "$(dirname "$0")/.." && pwd)
for n in 16384 32768 65536; do
  timeout -k 10 120 python $R/bench.py --no-cpu --no-configs --no-realtime --streams $n --interval-ms 20 --packets-per-step 2 --steps 300 > $out/b2b_$n.json 2>> $out/err.log || exit 1
done
for cs in 1 2 3; do for sub in 16384 32768 65536; do
  timeout -k 10 120 python $R/bench.py --paced --resident --streams 458752 --ticks 250 --sub-batch $sub --compute-streams $cs > $out/res_cs${cs}_sub$sub.json 2>> $out/err.log || exit 1
  timeout -k 10 120 python $R/bench.py --paced --streams 425984 --ticks 250 --sub-batch $sub --compute-streams $cs > $out/pcie_cs${cs}_sub$sub.json 2>> $out/err.log || exit 1
done; done
