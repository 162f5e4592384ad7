# round 4, third pass: full GPU suite (with the full-size oracle comparisons), then placement experiments on level 1
set -x
out=gpurun_out/${1:-r04_step3}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=8 > $out/tests.log 2>&1; echo "tests exit $?"; tail -14 $out/tests.log
timeout -k 10 300 python tools/bench_levels.py 256 20 --variants 2 --levels 3 --json $out/levels_a.json > $out/levels_a.log 2>&1
HYPRE_AMD_SPMV_RS_BAND_ROWS=-256 timeout -k 10 300 python tools/bench_levels.py 256 20 --variants 2 --levels 3 --json $out/levels_band256.json > $out/levels_band256.log 2>&1
HYPRE_AMD_SPMV_RS_BAND_ROWS=-128 timeout -k 10 300 python tools/bench_levels.py 256 20 --variants 2 --levels 3 --json $out/levels_band128.json > $out/levels_band128.log 2>&1
HYPRE_AMD_SPMV_RS_BAND_ROWS=-64 timeout -k 10 300 python tools/bench_levels.py 256 20 --variants 2 --levels 3 --json $out/levels_band64.json > $out/levels_band64.log 2>&1
timeout -k 10 300 python tools/bench_levels.py 256 20 --variants 2 --levels 3 --json $out/levels_b.json > $out/levels_b.log 2>&1
for f in a band256 band128 band64 b; do echo == $f; grep "V-cycle\|v2:0" $out/levels_$f.log; done
