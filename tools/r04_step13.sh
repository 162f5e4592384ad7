# round 4, step 13: the whole GPU suite with the one-workgroup tail and the wave-wide coarse solve; bench lines
set -x
out=gpurun_out/r04_step13
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1
rc=$?
tail -6 $out/tests.log
[ $rc -eq 0 ] || exit $rc
LEAK_ROUNDS=6 timeout -k 10 300 python tools/leak_check.py > $out/leak.log 2>&1; echo "leak exit $?"; tail -2 $out/leak.log
timeout -k 10 300 python bench.py > $out/bench_c2.json 2> $out/bench_c2.err; echo "c2 exit $?"
timeout -k 10 400 python bench.py --problem 27pt --relax 11 --no-cpu-baseline > $out/bench_c4.json 2> $out/bench_c4.err; echo "c4 exit $?"
timeout -k 10 300 python bench.py --problem difconv --mixed --no-cpu-baseline > $out/bench_c5.json 2> $out/bench_c5.err; echo "c5 exit $?"
python - <<'PY'
import json
for f in ("c2", "c4", "c5"):
    for l in open("gpurun_out/r04_step13/bench_%s.json" % f):
        if l.startswith('{"metric"'):
            d = json.loads(l)
            print(f, d["ms_per_step"], d["value"], d["roofline"]["frac"], d.get("ms_per_step_codes_off"), d["config"].get("coarse_tail_graph_nodes"), d["vcycle"].get("rel_max_err_vs_oracle", d["vcycle"]))
PY
