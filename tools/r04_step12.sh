set -x
out=gpurun_out/r04_step12
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_amg_gpu.py -m gpu -x -q -k "smallest or fusions or graph" > $out/tests.log 2>&1
rc=$?
tail -4 $out/tests.log
[ $rc -eq 0 ] || exit $rc
HYPRE_AMD_LIB=hypre_amd/lib/libhypre_amd_tailtiming.so timeout -k 10 300 python tools/experiments/tail_phases.py 256 2>&1 | tail -3
timeout -k 10 400 python tools/ab_row_slices.py 256 40 --toggle smalltail > $out/ab256.log 2>&1; tail -2 $out/ab256.log
timeout -k 10 300 python tools/ab_row_slices.py 128 100 --toggle smalltail > $out/ab128.log 2>&1; tail -2 $out/ab128.log
