# round 4, step 18: whole GPU suite with the LDS multicolour sweeps; the multicolour bench line
set -x
out=gpurun_out/r04_step18
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1
rc=$?
tail -5 $out/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --relax 21 --relax-up 22 --cpu-cycles 1 > $out/bench_mc256.json 2> $out/bench_mc256.err; echo "mc256 exit $?"
timeout -k 10 300 python bench.py --relax 21 --relax-up 22 --grid 128 --no-cpu-baseline > $out/bench_mc128.json 2> $out/bench_mc128.err; echo "mc128 exit $?"
python - <<'PY'
import json
for f in ("mc256", "mc128"):
    for l in open("gpurun_out/r04_step18/bench_%s.json" % f):
        if l.startswith('{"metric"'):
            d = json.loads(l)
            print(f, d["ms_per_step"], d["value"], d["config"].get("coarse_tail_graph_nodes"), d["pcg"]["iterations"], d["pcg"]["solve_ms"], (d.get("cpu_baseline") or {}).get("gpu_vs_cpu_cycle_rel_max_diff"))
PY
