# round 4, second pass: the persistent row-slice kernel with paired position words — tests of the form, then the per-level tables
set -x
out=gpurun_out/${1:-r04_step2}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_seq_matvec_gpu.py tests/test_amg_gpu.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests exit $?"; tail -5 $out/tests.log
timeout -k 10 300 python tools/bench_levels.py 256 20 --variants 2 --levels 4 --json $out/levels_rs1.json > $out/levels_rs1.log 2>&1; echo "levels rs1 exit $?"
HYPRE_AMD_SPMV_RS_PERSIST=0 timeout -k 10 300 python tools/bench_levels.py 256 20 --variants 2 --levels 4 --json $out/levels_np.json > $out/levels_np.log 2>&1; echo "levels np exit $?"
HYPRE_AMD_SPMV_RS_W=4 timeout -k 10 300 python tools/bench_levels.py 256 20 --variants 2 --levels 3 --json $out/levels_w4.json > $out/levels_w4.log 2>&1; echo "levels w4 exit $?"
HYPRE_AMD_SPMV_RS_PERSIST=768 timeout -k 10 300 python tools/bench_levels.py 256 20 --variants 2 --levels 3 --json $out/levels_p768.json > $out/levels_p768.log 2>&1; echo "levels p768 exit $?"
HYPRE_AMD_SPMV_RS_PERSIST=1536 timeout -k 10 300 python tools/bench_levels.py 256 20 --variants 2 --levels 3 --json $out/levels_p1536.json > $out/levels_p1536.log 2>&1; echo "levels p1536 exit $?"
for f in rs1 np w4 p768 p1536; do echo == $f; grep "V-cycle\|v2:0" $out/levels_$f.log; done
