#!/bin/bash
set -euo pipefail
mkdir -p gpurun_out/exp17 tools_dev/build
make -s -C wmix_amd/csrc OUT=$PWD/tools_dev/build/libwmix_amd_prof.so OBJDIR=$PWD/tools_dev/build/obj_prof EXTRA="-DWMX_NS_PROF -DWMX_AEC_PROF"
WMX_TOOL_LIB=tools_dev/build/libwmix_amd_prof.so python tools_dev/ns_prof.py > gpurun_out/exp17/ns_prof.txt 2>&1
WMX_TOOL_LIB=tools_dev/build/libwmix_amd_prof.so python tools_dev/aec_prof.py > gpurun_out/exp17/aec_prof.txt 2>&1
cat gpurun_out/exp17/ns_prof.txt gpurun_out/exp17/aec_prof.txt
