#!/bin/bash
# variant.sh -- ONE script for the developer A/B loop (it replaces the one-shot exp1.sh .. exp19.sh of rounds 3-4, whose findings
# are in DESIGN_HISTORY.md and under profiles/):
#
#   variant.sh build <name> "<flags>"        here (no GPU needed): tools_dev/build/lib_<name>.so with the Makefile's EXTRA="<flags>".
#                                            Timing experiments that change RESULTS (-DWMX_AEC_EXP=.., -DWMX_AEC_EXP_BARRIERS,
#                                            -DWMX_NS_EXP=1) need -DWMX_TIMING_ONLY_BUILD in the flags, or they do not compile
#                                            (wmix_amd/csrc/build_flags.h).  <flags> "" = a copy of the product build ("head").
#   variant.sh ab "<workloads>" <name> ...   on the GPU box: bench.py --no-cpu --steps 300 for each workload, the product library and
#                                            each named variant twice, interleaved: ms/step, dominant kernel ms, parity max LSB
#   variant.sh test <name> <pytest args>     on the GPU box: the GPU tests against a variant
#
# A variant library is refused by wmix_amd/_lib.py unless WMIX_AMD_ALLOW_VARIANT_BUILD=1: this script sets it, nothing else should.
set -euo pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
B=$R/tools_dev/build
cmd=${1:?build|ab|test}; shift
case $cmd in
  build)
    name=${1:?name}; flags=${2-}
    mkdir -p $B/obj_$name
    make -s -j 4 -C $R/wmix_amd/csrc OUT=$B/lib_$name.so OBJDIR=$B/obj_$name EXTRA="$flags"
    echo "built $B/lib_$name.so: $(WMIX_AMD_ALLOW_VARIANT_BUILD=1 WMIX_AMD_LIB=$B/lib_$name.so python -c 'from wmix_amd import _lib; print(_lib.build_info())')"
    ;;
  ab)
    wl=${1:?workloads}; shift
    export WMIX_AMD_ALLOW_VARIANT_BUILD=1
    for w in $wl; do for rep in 1 2; do for n in "" "$@"; do
      L=$R/wmix_amd/libwmix_amd.so; [ -n "$n" ] && L=$B/lib_$n.so
      WMIX_AMD_LIB=$L python $R/bench.py --workload $w --no-cpu --no-configs --steps 300 | python -c '
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2] or "product", d["build"], round(d["ms_per_step"], 4), (d["roofline"] or {}).get("avg_launch_ms"), (d.get("parity_checked") or {}).get("max_lsb"))' "$w" "$n"
    done; done; done
    ;;
  test)
    name=${1:?name}; shift
    WMIX_AMD_ALLOW_VARIANT_BUILD=1 WMIX_AMD_LIB=$B/lib_$name.so python -m pytest "$@"
    ;;
  *) echo "usage: variant.sh build|ab|test ..."; exit 2;;
esac
