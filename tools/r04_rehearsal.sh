# round 4: multi-rank GPU tests (ranks share the card) and rehearsals of bench.py --gpus 2 / 4 with the overlap diagnosis
set -x
out=gpurun_out/${1:-r04_rehearsal}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_dist_gpu.py tests/test_seq_matvec_gpu.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests exit $?"; tail -4 $out/tests.log
export HYPRE_AMD_BENCH_TRANSPORT=gloo
for cfg in "dev2 2" "dev4 4"; do
  set -- $cfg
  timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $2 --grid 128 --steps 10 --warmup 3 > $out/bench_rehearsal_$1.json 2> $out/bench_rehearsal_$1.err; echo "$1 exit $?"
done
unset HYPRE_AMD_BENCH_TRANSPORT
HYPRE_AMD_BENCH_SHARE_GPU=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --grid 96 --steps 5 --warmup 2 > $out/bench_rehearsal_rccl_fallback.json 2> $out/bench_rehearsal_rccl_fallback.err; echo "fallback exit $?"
tail -c 600 $out/bench_rehearsal_dev2.err
