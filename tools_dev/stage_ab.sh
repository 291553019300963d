#!/bin/bash
for L in "$@"; do
  echo "$L $(WMX_TOOL_LIB=$L python tools_dev/bench_lib.py --workload chain --no-cpu --steps 300 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); s=d["stage_ms"]; print(round(d["ms_per_step"],4), round(s["ns"],4), round(s["aec_near_kernel (timed region)"],4), round(s["agc"],4), round(s["vad"],4), d["parity_checked"]["max_lsb"])')"
done
