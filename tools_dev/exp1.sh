#!/bin/bash
# round-4 experiment 1: work-row stride (LDS bank conflicts) and FMA contraction of the near kernel, one GPU call
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp1; mkdir -p $O
B=tools_dev/build
bash tools_dev/ab.sh chain $B/lib_head.so wmix_amd/libwmix_amd.so $B/lib_fas132.so $B/lib_fma.so $B/lib_head.so wmix_amd/libwmix_amd.so > $O/ab.txt 2>&1
cat $O/ab.txt
python -m pytest tests/test_aec_gpu.py -q -m gpu > $O/aec_default.txt 2>&1; tail -3 $O/aec_default.txt
WMIX_AMD_LIB=$B/lib_fma.so python -m pytest tests/test_aec_gpu.py tests/test_vs_reference_gpu.py -q -m gpu > $O/aec_fma.txt 2>&1; tail -15 $O/aec_fma.txt
for v in head default; do
  L=wmix_amd/libwmix_amd.so; [ $v = head ] && L=$B/lib_head.so
  WMIX_AMD_LIB=$R/$L bash tools_dev/pmc_pass.sh lds_$v SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS 2>&1 | grep aec_near | tee $O/pmc_lds_$v.txt
done
