#!/bin/bash
set -uo pipefail
python -m pytest tests/test_vadagc_gpu.py tests/test_configs_gpu.py tests/test_cadence_gpu.py tests/test_extremes_gpu.py tests/test_edges_gpu.py -q -m gpu 2>&1 | tail -2
one() { python bench.py --no-cpu --steps 200 "$@" | python -c "import sys,json; d=json.load(sys.stdin); print(round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stage_ms'].items() if k in ('agc','vad','ns','mix')}, (d.get('parity_checked') or {}).get('max_lsb'))"; }
for rep in 1 2; do
  for L in wmix_amd/libwmix_amd.so tools_dev/build/lib_prev.so; do
    echo "$L"; WMIX_AMD_LIB=$L one; WMIX_AMD_LIB=$L one --interval-ms 20 --packets-per-step 2; WMIX_AMD_LIB=$L one --workload ns_agc_mix_32k; WMIX_AMD_LIB=$L one --packets-per-step 4
  done
done
