# round-2 measurement pass on the GPU box: bench lines of the BASELINE configurations + kernel statistics
set -x
out=gpurun_out/r02_measure
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py > $out/bench_c2.json 2> $out/bench_c2.err; echo "c2 exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof_c2 -o b --output-format csv -- python3 bench.py --no-cpu-baseline > $out/prof_c2.log 2>&1; echo "prof exit $?"
timeout -k 10 400 python bench.py --problem 27pt --relax 11 > $out/bench_c4.json 2> $out/bench_c4.err; echo "c4 exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof_c4 -o b --output-format csv -- python3 bench.py --problem 27pt --relax 11 --no-cpu-baseline > $out/prof_c4.log 2>&1
timeout -k 10 300 python bench.py --problem difconv --mixed > $out/bench_c5.json 2> $out/bench_c5.err; echo "c5 exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof_c5 -o b --output-format csv -- python3 bench.py --problem difconv --mixed --no-cpu-baseline > $out/prof_c5.log 2>&1
timeout -k 10 300 python bench.py --grid 128 --relax 21 --relax-up 22 --cpu-cycles 1 > $out/bench_mc128.json 2> $out/bench_mc128.err; echo "mc128 exit $?"
timeout -k 10 300 python bench.py --relax 21 --relax-up 22 --cpu-cycles 1 > $out/bench_mc256.json 2> $out/bench_mc256.err; echo "mc256 exit $?"
timeout -k 10 300 python bench.py --grid 128 --relax 13 --relax-up 14 --cpu-cycles 1 > $out/bench_gs128.json 2> $out/bench_gs128.err; echo "gs128 exit $?"
HYPRE_AMD_BENCH_TRANSPORT=gloo HYPRE_AMD_BENCH_SHARE_GPU=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --grid 128 --steps 10 --warmup 3 > $out/bench_2rank_rehearsal.json 2> $out/bench_2rank_rehearsal.err; echo "2rank exit $?"
tail -2 $out/bench_2rank_rehearsal.err
# HYPRE_BoomerAMGSetup alone (matrix handed over in device / host memory), and its kernels
HYPRE_AMD_SETUP_TIMING=1 timeout -k 10 300 python tools/setup_time.py 256 device 3 > $out/setup_device.log 2>&1; echo "setup exit $?"
timeout -k 10 300 python tools/setup_time.py 256 host 2 > $out/setup_host.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof_setup -o s --output-format csv -- python3 tools/setup_time.py 256 device 1 > $out/prof_setup.log 2>&1
ls $out
