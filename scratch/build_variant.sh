#!/bin/bash
# a variant build of the library: ONE source recompiled with extra flags, the other objects
# taken from the in-tree build.   scratch/build_variant.sh NAME SOURCE.hip FLAGS...   -> scratch/ab/lib_NAME.so
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../mdlm_amd/csrc"
make -s -j8 > /dev/null
mkdir -p ../../scratch/ab
o=../../scratch/ab/${src%.hip}_$name.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c $src -o $o
others=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $o -o ../../scratch/ab/lib_$name.so
ls -la ../../scratch/ab/lib_$name.so
