#!/bin/bash
# A/B of library builds on the GPU box: per-level timings of every copy named, twice, interleaved.
#   tools/ab_levels.sh <outdir> <levels> <lib-suffix>...      ("" = the default library)
out=$1; levels=$2; shift 2
mkdir -p $out
for rep in 1 2; do
  for v in "$@"; do
    lib=$PWD/hypre_amd/lib/libhypre_amd${v:+_$v}.so
    HYPRE_AMD_LIB=$lib timeout -k 10 200 python tools/bench_levels.py 256 20 --variants 2 --levels $levels > $out/levels_${v:-default}_$rep.log 2>&1 || exit 1
  done
done
for v in "$@"; do for rep in 1 2; do echo "== ${v:-default} $rep"; grep -h "V-cycle\|v2:0" $out/levels_${v:-default}_$rep.log | cut -c1-200; done; done
