#!/bin/bash
# Recipe behind profiles/rNN/chain_*: run on the GPU box as
#   gpurun -- 'bash profiles/tools/profile_chain.sh r01'
# Three separate rocprofv3 runs of the same bench command (kernel trace + stats; FETCH_SIZE pass; WRITE_SIZE
# pass -- PMC passes carry --kernel-trace only, as the pool requires), summarised into gpurun_out/<round>/.
set -u
ROUND=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$ROUND
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
STEPS=40
CMD="python3 $R/bench.py --no-cpu --steps $STEPS --warmup 8"   # bench.py's defaults
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o chain --output-format csv -- $CMD > "$OUT/bench_under_stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o chain --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o chain --output-format csv -- $CMD > /dev/null 2>&1
# where the waves' time goes (one SQ pass; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES)
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES \
  -d "$OUT/pmc_sq" -o chain --output-format csv -- $CMD > /dev/null 2>&1
cd "$R"
python3 profiles/tools/summarise.py "$OUT" 65536 $STEPS
python3 bench.py --steps 40 --warmup 8 > "$OUT/bench_line.json" 2> "$OUT/bench_stderr.log"
tail -c 1500 "$OUT/bench_line.json"
