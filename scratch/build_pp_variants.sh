#!/bin/bash
# release-like builds of the library with compile-time timing experiments in gemm_pp.hip:
#   scratch/ab/lib_pp_<name>.so for name in the arguments, e.g. "nl:-DPP_NO_LOADS ns:-DPP_NO_STORES"
set -e
cd "$(dirname "$0")/../mdlm_amd/csrc"
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  d=/tmp/ppv_$name; mkdir -p $d
  cp *.o $d/
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 ${flags//,/ } -c gemm_pp.hip -o $d/gemm_pp.o &
done
wait
for spec in "$@"; do
  name=${spec%%:*}; d=/tmp/ppv_$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $d/*.o -o ../../scratch/ab/lib_pp_$name.so
done
ls ../../scratch/ab/
