# slice form of coded stencils: parity tests, then the per-level table and the bench lines with the form on and off
set -x
out=gpurun_out/${1:-r03_slice}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_seq_matvec_gpu.py tests/test_amg_gpu.py tests/test_multicolor_gpu.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests exit $?"; tail -15 $out/tests.log
timeout -k 10 300 python tools/bench_levels.py 256 20 --variants 2 --levels 1 > $out/levels_7pt_on.log 2>&1
HYPRE_AMD_SPMV_SLICE_FORM=0 timeout -k 10 300 python tools/bench_levels.py 256 20 --variants 2 --levels 1 > $out/levels_7pt_off.log 2>&1
timeout -k 10 300 python tools/bench_levels.py 256 20 --variants 2 --levels 1 --problem 27pt --relax 11 > $out/levels_27pt_on.log 2>&1
HYPRE_AMD_SPMV_SLICE_FORM=0 timeout -k 10 300 python tools/bench_levels.py 256 20 --variants 2 --levels 1 --problem 27pt --relax 11 > $out/levels_27pt_off.log 2>&1
grep -h "V-cycle\|A x" $out/levels_7pt_on.log $out/levels_7pt_off.log $out/levels_27pt_on.log $out/levels_27pt_off.log
