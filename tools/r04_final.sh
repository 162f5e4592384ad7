# final pass of round 4 on one box: GPU suite, PMC passes of the level kernels (final kernel source), bench lines, level tables
set -x
out=gpurun_out/${1:-r04_final}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests exit $?"; tail -3 $out/tests.log
bash tools/pmc_levels.sh r04_levels_rs 2 > $out/pmc.log 2>&1; python tools/pmc_levels_summary.py r04_levels_rs > $out/pmc_summary.txt 2>&1; cp profiles/r04_levels_rs_summary.json $out/
timeout -k 10 300 python bench.py > $out/bench_c2.json 2> $out/bench_c2.err; echo "c2 exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof_c2 -o b --output-format csv -- python3 bench.py --no-cpu-baseline > $out/prof_c2.log 2>&1; echo "prof exit $?"
timeout -k 10 400 python bench.py --problem 27pt --relax 11 > $out/bench_c4.json 2> $out/bench_c4.err; echo "c4 exit $?"
timeout -k 10 300 python bench.py --problem difconv --mixed > $out/bench_c5.json 2> $out/bench_c5.err; echo "c5 exit $?"
timeout -k 10 300 python bench.py --relax 21 --relax-up 22 --cpu-cycles 1 > $out/bench_mc256.json 2> $out/bench_mc256.err; echo "mc256 exit $?"
HYPRE_AMD_SPMV_ROW_SLICES=0 timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench_c2_norowslices.json 2> $out/bench_c2_norowslices.err; echo "c2 nors exit $?"
timeout -k 10 300 python tools/bench_levels.py 256 20 --variants 2 --json $out/levels_7pt.json > $out/levels_7pt.log 2>&1; echo "levels 7pt exit $?"
timeout -k 10 300 python tools/bench_levels.py 256 20 --variants 2 --problem 27pt --relax 11 --json $out/levels_27pt.json > $out/levels_27pt.log 2>&1; echo "levels 27pt exit $?"
grep "V-cycle\|v2:0\|plan form" $out/levels_7pt.log | head -16
