# round 4, step 10: the multivector goldens (DS-PCG, vector.saved) on the device, then the whole GPU suite
set -x
out=gpurun_out/r04_step10
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_ij_cli.py tests/test_dist_gpu.py tests/test_seq_matvec_gpu.py -m gpu -x -q -k "vector or ds_pcg or multivector" > $out/tests_mv.log 2>&1
rc=$?
tail -15 $out/tests_mv.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1
rc=$?
tail -4 $out/tests.log
exit $rc
