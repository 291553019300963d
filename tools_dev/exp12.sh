#!/bin/bash
# per-phase split of nsx_kernel / aecm_near_kernel on the current tree
set -euo pipefail
mkdir -p gpurun_out/exp12 tools_dev/build
make -s -C wmix_amd/csrc OUT=$PWD/tools_dev/build/libwmix_amd_prof.so OBJDIR=$PWD/tools_dev/build/obj_prof EXTRA="-DWMX_NSX_PROF -DWMX_AECM_PROF"
WMX_TOOL_LIB=tools_dev/build/libwmix_amd_prof.so python tools_dev/nsx_prof.py > gpurun_out/exp12/nsx_prof.txt 2>&1
WMX_TOOL_LIB=tools_dev/build/libwmix_amd_prof.so python tools_dev/aecm_prof.py > gpurun_out/exp12/aecm_prof.txt 2>&1
cat gpurun_out/exp12/nsx_prof.txt gpurun_out/exp12/aecm_prof.txt
