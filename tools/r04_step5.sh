# round 4, fifth pass: the restriction that also starts the coarse sweep — tests, then A/B of the cycle in one process
set -x
out=gpurun_out/${1:-r04_step5}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_amg_gpu.py tests/test_bench_class_gpu.py tests/test_full_size_gpu.py tests/test_multicolor_gpu.py tests/test_ij_cli.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests exit $?"; tail -4 $out/tests.log
timeout -k 10 300 python tools/ab_row_slices.py 256 30 --toggle fusion > $out/ab_fusion_c2.txt 2>&1; tail -2 $out/ab_fusion_c2.txt
timeout -k 10 300 python tools/ab_row_slices.py 256 30 --toggle fusion --codes 0 > $out/ab_fusion_c2_codes_off.txt 2>&1; tail -2 $out/ab_fusion_c2_codes_off.txt
timeout -k 10 300 python tools/ab_row_slices.py 256 30 --toggle fusion --problem difconv > $out/ab_fusion_c5.txt 2>&1; tail -2 $out/ab_fusion_c5.txt
timeout -k 10 300 python tools/ab_row_slices.py 128 50 --toggle fusion > $out/ab_fusion_128.txt 2>&1; tail -2 $out/ab_fusion_128.txt
