#!/bin/bash
# ablation build of the library (timing switches in the GEMM kernels): scratch/ab/lib_abl.so
set -e
cd "$(dirname "$0")/../mdlm_amd/csrc"
mkdir -p ../../scratch/ab/obj
for f in *.hip; do
  o=../../scratch/ab/obj/${f%.hip}.o
  if [ ! -f $o ] || [ $f -nt $o ] || [ gemm_common.h -nt $o ] || [ common.h -nt $o ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMDLM_GEMM_ABLATION -c $f -o $o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ../../scratch/ab/obj/*.o -o ../../scratch/ab/lib_abl.so
ls -la ../../scratch/ab/lib_abl.so
