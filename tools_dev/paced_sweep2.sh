#!/bin/bash
# paced_sweep2.sh <outdir> -- on the GPU box: staggered release (P groups tick / P apart), resident and from host memory
set -uo pipefail
out=${1:?outdir}; mkdir -p $out
R=$(cd "$(dirname "$0")/.." && pwd)
for P in 4 8; do
  timeout -k 10 300 python $R/bench.py --paced --resident --phases $P --sub-batch 65536 --ticks 300 --paced-search 524288,557056,589824,622592 > $out/res_P$P.json 2>> $out/err.log || exit 1
  timeout -k 10 300 python $R/bench.py --paced --phases $P --sub-batch 32768 --ticks 300 --paced-search 491520,524288,557056,589824 > $out/pcie_P$P.json 2>> $out/err.log || exit 1
done
