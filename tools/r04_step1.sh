# round 4, first pass: GPU tests, then the row-slice form on the benchmark hierarchy — per-level tables with the form off
# (round 3's kernels), on (default width) and at forced widths; one bench line
set -x
out=gpurun_out/${1:-r04_step1}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests exit $?"; tail -5 $out/tests.log
HYPRE_AMD_SPMV_ROW_SLICES=0 timeout -k 10 300 python tools/bench_levels.py 256 20 --variants 2 --levels 4 --json $out/levels_rs0.json > $out/levels_rs0.log 2>&1; echo "levels rs0 exit $?"
HYPRE_AMD_PLAN_VERBOSE=1 timeout -k 10 300 python tools/bench_levels.py 256 20 --variants 2 --levels 4 --json $out/levels_rs1.json > $out/levels_rs1.log 2>&1; echo "levels rs1 exit $?"
for w in 2 4 8; do
  HYPRE_AMD_SPMV_RS_W=$w timeout -k 10 300 python tools/bench_levels.py 256 20 --variants 2 --levels 4 --json $out/levels_rsw$w.json > $out/levels_rsw$w.log 2>&1; echo "levels w$w exit $?"
done
timeout -k 10 300 python bench.py > $out/bench_c2.json 2> $out/bench_c2.err; echo "c2 exit $?"
cat $out/levels_rs0.log | head -40
cat $out/levels_rs1.log | head -60
