# build a variant of the library with extra compiler flags for spmv_kernels.hip only (run in the build container):
#   bash tools/experiments/build_variant.sh timing -DXS_TIMING=1   ->   hypre_amd/lib/libhypre_amd_timing.so
# load it with HYPRE_AMD_LIB=hypre_amd/lib/libhypre_amd_<name>.so
set -e
name=$1; shift
cd "$(dirname "$0")/../.."
python -m hypre_amd.build > /dev/null
mkdir -p hypre_amd/lib/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fopenmp -Iinclude -Ihypre_amd/csrc "$@" -x hip -c hypre_amd/csrc/spmv_kernels.hip -o hypre_amd/lib/exp/spmv_kernels_$name.o
objs=$(ls hypre_amd/lib/obj/*.o | grep -v spmv_kernels.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fopenmp -o hypre_amd/lib/libhypre_amd_$name.so $objs hypre_amd/lib/exp/spmv_kernels_$name.o -lrccl
echo built hypre_amd/lib/libhypre_amd_$name.so
