#!/bin/bash
# multi-rank rehearsals on the one-GPU box: ranks share the card, halo bytes over gloo (stream-staged transport)
set -o pipefail
out=gpurun_out/${1:-r3b}
mkdir -p $out
export HYPRE_AMD_BENCH_TRANSPORT=gloo HYPRE_AMD_SETUP_TIMING=1
run() {  # name ranks grid extra-env...
  name=$1; ranks=$2; grid=$3; shift 3
  env "$@" timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $ranks --master-addr 127.0.0.1 --master-port 29517 \
      bench.py --gpus $ranks --grid $grid --steps 10 --warmup 3 $BENCH_ARGS > $out/$name.json 2> $out/$name.err
  echo "$name rc=$?"; tail -c 600 $out/$name.json; echo
}
run dev2_128 2 128
run host2_128 2 128 HYPRE_AMD_SETUP_DEVICE_DIST=0
run dev4_128 4 128
