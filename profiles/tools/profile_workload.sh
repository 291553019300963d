#!/bin/bash
# Recipe behind profiles/rNN/<workload>_*: run on the GPU box as
#   gpurun -- 'bash profiles/tools/profile_workload.sh r02 chain 65536'
#   gpurun -- 'bash profiles/tools/profile_workload.sh r02 nsx 65536'  ...
# Separate rocprofv3 runs of the same bench command, as the pool requires for PMC passes (--kernel-trace only beside
# --pmc): kernel trace + stats; FETCH_SIZE; WRITE_SIZE; where the waves' time goes (SQ_WAIT* / SQ_ACTIVE_*); instruction
# counts; LDS bank conflicts + occupancy.  Summaries land in gpurun_out/<round>/ -- copy the ones to be judged into
# profiles/<round>/.
set -u
ROUND=${1:-r03}; WL=${2:-chain}; NFR=${3:-65536}; shift 3 || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$ROUND/$WL
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
STEPS=40
CMD="python3 $R/bench.py --no-cpu --no-configs --workload $WL --steps $STEPS --warmup 8 $*"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o p --output-format csv -- $CMD > "$OUT/bench_under_stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o p --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o p --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES \
  -d "$OUT/pmc_sq" -o p --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM \
  -d "$OUT/pmc_sq_insts" -o p --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES \
  -d "$OUT/pmc_sq_lds" -o p --output-format csv -- $CMD > /dev/null 2>&1
# lane utilisation of the vector ALU (round-4 VERDICT "next" 9): thread-cycles against instruction-cycles x 64 lanes
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
  -d "$OUT/pmc_sq_lanes" -o p --output-format csv -- $CMD > /dev/null 2>&1
cd "$R"
python3 profiles/tools/summarise.py "$OUT" "$NFR" $STEPS "$WL"
# the raw traces are tens of MB per workload and gpurun brings back at most 64 MiB: keep the summaries only
[ -n "${KEEP_RAW:-}" ] || rm -rf "$OUT"/stats "$OUT"/pmc_*
ls "$OUT"/*.csv "$OUT"/*.json 2>/dev/null
