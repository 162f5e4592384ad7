#!/bin/bash
# AddressSanitizer pass over the HOST side of the library (setup, comm packages, bindings) on a CPU box:
# builds a copy of the library with the host code instrumented (-Xarch_host -fsanitize=address; device code
# untouched), then runs the CPU test-suite against it.  GPU sanitizers are not available on the pool.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
W=${TMPDIR:-/tmp}/hypre_amd_asan
rm -rf "$W" && mkdir -p "$W" && cp -r "$ROOT/hypre_amd" "$ROOT/include" "$W/"
sed -i 's|FLAGS = \["--offload-arch=gfx950", "-O3"|FLAGS = ["-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fno-omit-frame-pointer", "-g", "--offload-arch=gfx950", "-O1"|' "$W/hypre_amd/build.py"
sed -i 's|"-shared", "-fPIC", "-fopenmp"|"-shared", "-fPIC", "-fopenmp", "-fsanitize=address", "-shared-libasan"|' "$W/hypre_amd/build.py"
(cd "$W" && python hypre_amd/build.py --force > /dev/null 2>&1)
ASAN_SO=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd "$ROOT"
HYPRE_AMD_LIB="$W/hypre_amd/lib/libhypre_amd.so" LD_PRELOAD="$ASAN_SO" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
  python -m pytest tests -x -q -m "not gpu" "$@"
