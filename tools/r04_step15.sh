# round 4, step 15: bench.py --gpus 2 / 4 rehearsals (ranks share the card) with the library as it stands
set -x
out=gpurun_out/r04_step15
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HYPRE_AMD_BENCH_TRANSPORT=gloo
for cfg in "dev2 2 --grid 128" "dev4 4 --grid 128" "c4_dev2 2 --grid 96 --problem 27pt --relax 11"; do
  set -- $cfg
  name=$1; np=$2; shift 2
  timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $np --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $np --steps 10 --warmup 3 "$@" > $out/bench_rehearsal_$name.json 2> $out/bench_rehearsal_$name.err; echo "$name exit $?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_step15/bench_rehearsal_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{"metric"')][0]); c=d['config']; o=c['overlap']
        print(f.split('/')[-1], 'ms/step %.3f' % d['ms_per_step'], c['transport'], '| exposed %.0f transfer %.0f host %.0f us | single %.3f ms | parity %s' % (o['exposed_us_per_cycle'], o['transfer_us_per_cycle'], o['host_in_transport_us_per_cycle'], o.get('single_rank_ms_per_step_same_block',-1), (d.get('cpu_baseline') or {}).get('gpu_vs_cpu_cycle_rel_max_diff')))
    except Exception as e: print(f, 'ERR', e)
PY
