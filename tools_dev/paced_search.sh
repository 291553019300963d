#!/bin/bash
# paced_search.sh <outdir> -- on the GPU box: S_max searches with the C host (1500 ticks per stream count, stop at the first miss)
set -uo pipefail
out=${1:?outdir}; mkdir -p $out
R=$(cd "$(dirname "$0")/.." && pwd)
H="python $R/tools_dev/paced_host.py --ticks 1500 --stop-at-miss"
$H --streams 622592,655360 --phases 4 --sub 65536 --out $out/pcm16k_p4.jsonl > /dev/null || exit 1
$H --kind rtp8k --streams 655360,786432,917504,1048576 --sub 65536 --out $out/rtp8k_p1.jsonl > /dev/null || exit 1
$H --kind rtp8k --streams 917504,1048576,1179648,1310720 --phases 4 --sub 65536 --out $out/rtp8k_p4.jsonl > /dev/null || exit 1
$H --tick-ms 10 --streams 163840,196608,229376,262144 --out $out/pcm16k_10ms_p1.jsonl > /dev/null || exit 1
$H --tick-ms 10 --streams 229376,262144,294912,327680 --phases 4 --sub 65536 --out $out/pcm16k_10ms_p4.jsonl > /dev/null || exit 1
