out=gpurun_out/r3m; mkdir -p $out
export HYPRE_AMD_BENCH_TRANSPORT=gloo
HYPRE_AMD_BENCH_FAKE_SETUP_ERROR=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --grid 96 --steps 5 --warmup 2 --no-cpu-baseline > $out/fallback.json 2> $out/fallback.err
echo rc=$?
grep "repeating" $out/fallback.err | head -3
python -c "
import json; d=json.loads(open('$out/fallback.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['setup_path'], d['config']['setup_seconds'], d['pcg'].get('iterations'))"
