# value codes (one byte per entry for matrices with <= 256 distinct values): parity tests, then the benchmark cycle and the
# per-level table with the codes on and off
set -x
out=gpurun_out/${1:-r03_codes}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_seq_matvec_gpu.py tests/test_amg_gpu.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests exit $?"; tail -3 $out/tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench_c2_on.json 2> $out/bench_c2_on.err; echo "c2 on exit $?"
HYPRE_AMD_SPMV_VALUE_CODES=0 timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench_c2_off.json 2> $out/bench_c2_off.err; echo "c2 off exit $?"
timeout -k 10 300 python tools/bench_levels.py 256 20 --variants 2 --levels 2 > $out/levels_7pt_on.log 2>&1; echo "levels exit $?"
timeout -k 10 400 python bench.py --problem 27pt --relax 11 --no-cpu-baseline > $out/bench_c4_on.json 2> $out/bench_c4_on.err; echo "c4 exit $?"
timeout -k 10 300 python bench.py --problem difconv --mixed --no-cpu-baseline > $out/bench_c5_on.json 2> $out/bench_c5_on.err; echo "c5 exit $?"
grep -h ms_per_step $out/bench_*.json | cut -c1-220
cat $out/levels_7pt_on.log
