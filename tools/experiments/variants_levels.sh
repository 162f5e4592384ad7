# per-level table (first three levels) and the cycle for library variants built by tools/experiments/build_variant.sh
out=gpurun_out/${1:-r03_variants}; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/hypre_amd/lib/libhypre_amd_$v.so
  [ "$v" = default ] && lib=$GRAFT_REPO_ROOT/hypre_amd/lib/libhypre_amd.so
  HYPRE_AMD_LIB=$lib timeout -k 10 300 python tools/bench_levels.py 256 20 --variants 2 --levels 3 > $out/levels_$v.log 2>&1; echo "$v exit $?"
  grep -h "V-cycle\|A x" $out/levels_$v.log
done
