# round 4, step 14: two-stage Gauss-Seidel in the one-workgroup tail: tests, A/B on the 27-point problem, bench C4
set -x
out=gpurun_out/r04_step14
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_amg_gpu.py tests/test_bench_class_gpu.py -m gpu -x -q > $out/tests.log 2>&1
rc=$?
tail -12 $out/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python tools/ab_row_slices.py 256 30 --toggle smalltail --problem 27pt --relax 11 > $out/ab27.log 2>&1; tail -2 $out/ab27.log
timeout -k 10 300 python tools/ab_row_slices.py 128 60 --toggle smalltail --problem 27pt --relax 11 > $out/ab27_128.log 2>&1; tail -2 $out/ab27_128.log
