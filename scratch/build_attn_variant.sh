#!/bin/bash
# build_attn_variant.sh NAME "-DFLAGS": scratch/ab/lib_NAME.so with attn_fwd.hip rebuilt under FLAGS
set -e
cd /root/repo/mdlm_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $2 -c attn_fwd.hip -o /tmp/attn_$1.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "attn_fwd_pipe" | grep -E "VGPRs:|Spill" | tr '\n' ' '
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v attn_fwd.o) /tmp/attn_$1.o -o /root/repo/scratch/ab/lib_$1.so
echo " -> lib_$1.so"
