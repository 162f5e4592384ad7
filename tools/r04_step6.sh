# round 4, sixth pass: the two-stage sweep in two or three passes — tests, then A/B of the C4 cycle in one process
set -x
out=gpurun_out/${1:-r04_step6}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_amg_gpu.py tests/test_bench_class_gpu.py tests/test_full_size_gpu.py tests/test_dist_gpu.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests exit $?"; tail -4 $out/tests.log
timeout -k 10 400 python tools/ab_row_slices.py 256 20 --toggle fusion --problem 27pt --relax 11 > $out/ab_fusion_c4.txt 2>&1; tail -2 $out/ab_fusion_c4.txt
timeout -k 10 400 python tools/ab_row_slices.py 256 20 --toggle fusion --problem 27pt --relax 11 --codes 0 > $out/ab_fusion_c4_codes_off.txt 2>&1; tail -2 $out/ab_fusion_c4_codes_off.txt
timeout -k 10 300 python tools/ab_row_slices.py 256 30 --toggle fusion --relax 11 > $out/ab_fusion_7pt_tsgs.txt 2>&1; tail -2 $out/ab_fusion_7pt_tsgs.txt
