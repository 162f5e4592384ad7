#!/bin/bash
# Build a second copy of the library with extra -D flags on ONE source (A/B runs of a kernel on the GPU box:
# HYPRE_AMD_LIB=<copy> python tools/bench_levels.py ...).   tools/build_variant_lib.sh <name> <source> <flags...>
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
obj=/tmp/variant_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fopenmp -I$root/include -I$root/hypre_amd/csrc "$@" -x hip -c $root/hypre_amd/csrc/$src -o $obj
objs=$(ls $root/hypre_amd/lib/obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fopenmp -o $root/hypre_amd/lib/libhypre_amd_$name.so $objs $obj -lrccl
echo built $root/hypre_amd/lib/libhypre_amd_$name.so
