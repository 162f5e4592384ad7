# final measurement pass of round 3 (value codes + slice form): bench lines, kernel statistics, per-level tables, multi-rank
# rehearsals, setup timing
set -x
out=gpurun_out/${1:-r03_measure3}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py > $out/bench_c2.json 2> $out/bench_c2.err; echo "c2 exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof_c2 -o b --output-format csv -- python3 bench.py --no-cpu-baseline > $out/prof_c2.log 2>&1; echo "prof exit $?"
timeout -k 10 400 python bench.py --problem 27pt --relax 11 > $out/bench_c4.json 2> $out/bench_c4.err; echo "c4 exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof_c4 -o b --output-format csv -- python3 bench.py --problem 27pt --relax 11 --no-cpu-baseline > $out/prof_c4.log 2>&1
timeout -k 10 300 python bench.py --problem difconv --mixed > $out/bench_c5.json 2> $out/bench_c5.err; echo "c5 exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof_c5 -o b --output-format csv -- python3 bench.py --problem difconv --mixed --no-cpu-baseline > $out/prof_c5.log 2>&1
timeout -k 10 300 python bench.py --relax 21 --relax-up 22 --cpu-cycles 1 > $out/bench_mc256.json 2> $out/bench_mc256.err; echo "mc256 exit $?"
HYPRE_AMD_SPMV_SLICE_FORM=0 timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench_c2_noslice.json 2> $out/bench_c2_noslice.err; echo "c2 noslice exit $?"
HYPRE_AMD_SPMV_VALUE_CODES=0 timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench_c2_nocodes.json 2> $out/bench_c2_nocodes.err; echo "c2 nocodes exit $?"
HYPRE_AMD_SPMV_VALUE_CODES=0 timeout -k 10 400 python bench.py --problem 27pt --relax 11 --no-cpu-baseline > $out/bench_c4_nocodes.json 2> $out/bench_c4_nocodes.err; echo "c4 nocodes exit $?"
timeout -k 10 300 python tools/bench_levels.py 256 20 --variants 2 --json $out/levels_7pt.json > $out/levels_7pt.log 2>&1; echo "levels 7pt exit $?"
timeout -k 10 300 python tools/bench_levels.py 256 20 --variants 2 --problem 27pt --relax 11 --json $out/levels_27pt.json > $out/levels_27pt.log 2>&1; echo "levels 27pt exit $?"
export HYPRE_AMD_BENCH_TRANSPORT=gloo
for cfg in "dev2 2 HYPRE_AMD_SETUP_DEVICE_DIST=1" "dev4 4 HYPRE_AMD_SETUP_DEVICE_DIST=1"; do
  set -- $cfg
  env $3 HYPRE_AMD_SETUP_TIMING=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $2 --grid 128 --steps 10 --warmup 3 > $out/bench_rehearsal_$1.json 2> $out/bench_rehearsal_$1.err; echo "$1 exit $?"
done
unset HYPRE_AMD_BENCH_TRANSPORT
HYPRE_AMD_SETUP_TIMING=1 timeout -k 10 300 python tools/setup_time.py 256 device 3 > $out/setup_device.log 2>&1; echo "setup exit $?"
ls $out
