#!/bin/bash
# sanitize_cpu.sh [pytest args] -- the CPU test suite with everything that runs on the host under AddressSanitizer +
# UndefinedBehaviorSanitizer, -fno-sanitize-recover (any finding aborts the run and shows as a failed step):
#   1. oracle/orc_*.c, the C restatement, as oracle/build/liboracle_san.so (oracle/Makefile `san`), loaded by every oracle
#      test through oracle/loader.py (WMIX_ORACLE_SAN=1; the ASan runtime is preloaded into the interpreter);
#   2. the library's host-side control planes -- aec_ctl.h, aecm_ctl.h, the AGC gain-table recipe, the zoom / load schedules
#      of the mixer -- compiled WITHOUT the HIP runtime into tools_dev/san/host_ctl_san and driven over their whole
#      argument ranges (tools_dev/san/host_ctl_san.cpp).
# SURVEY section 5 asks for sanitizers on the CPU code; GPU AddressSanitizer is not available on this pool.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
make -s -C oracle san
make -s -C tools_dev/san
echo "== host control planes under ASan + UBSan"
ASAN_OPTIONS=detect_leaks=1 ./tools_dev/san/host_ctl_san
echo "== CPU suite against the sanitized oracle"
ASAN=$(gcc -print-file-name=libasan.so)
# python itself is not instrumented: leak reports from the interpreter are noise
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 WMIX_ORACLE_SAN=1 python -m pytest tests -q -x -m "not gpu" -p no:cacheprovider "$@"
