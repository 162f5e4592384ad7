#!/bin/bash
# A/B the SpMV tuning knobs on the fine-level matrix: tools/ab_spmv.sh [n]
n=${1:-256}
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" timeout -k 10 120 python tools/bench_spmv.py $n 50 2>&1 | grep -v generate; }
run HYPRE_AMD_SPMV_XCD=0
run HYPRE_AMD_SPMV_XCD=7
run HYPRE_AMD_SPMV_XCD=14
run HYPRE_AMD_SPMV_XCD=28
run HYPRE_AMD_SPMV_XCD=56
run HYPRE_AMD_SPMV_XCD=16
run HYPRE_AMD_SPMV_XCD=-1
run HYPRE_AMD_SPMV_GT=0
run SPMV_LOCAL_COLS=1
