# larger multi-rank rehearsals (ranks share the card, gloo transport): C4 / C5 per-GPU shapes at 128^3, C3 at 2 x 256^3
out=gpurun_out/${1:-r03_rehearsal2}
mkdir -p $out
export HYPRE_AMD_BENCH_TRANSPORT=gloo HYPRE_AMD_SETUP_TIMING=1
run() { name=$1; ranks=$2; shift 2
  timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $ranks --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus $ranks --steps 10 --warmup 3 --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err
  echo "$name rc=$?"; }
run c4_4x128 4 --grid 128 --problem 27pt --relax 11
run c5_4x128 4 --grid 128 --problem difconv --mixed
run c3_2x256 2 --grid 256
python - <<PY
import json
for f in ("c4_4x128", "c5_4x128", "c3_2x256"):
    try:
        d = json.loads(open("$out/%s.json" % f).read().strip().splitlines()[-1])
        c = d["config"]
        print(f, "ms/cycle %.3f" % d["ms_per_step"], "setup %.3f" % c["setup_seconds"], "levels", c["levels"], "replicated from", c["replicated_from_level"],
              "exchanges", c["halo_exchanges_per_cycle"], "bytes", c["halo_bytes_sent_per_cycle"], "pcg", d["pcg"].get("iterations"), d["pcg"].get("final_rel_resid"))
    except Exception as e:
        print(f, "FAILED", e)
PY
