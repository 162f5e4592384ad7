# round 4, step 11: the one-workgroup tail — its test, the AMG tests, the bench line with and without it
set -x
out=gpurun_out/r04_step11
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_amg_gpu.py -m gpu -x -q > $out/tests_amg.log 2>&1
rc=$?
tail -15 $out/tests_amg.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench_on.json 2> $out/bench_on.err && HYPRE_AMD_SMALL_TAIL=0 timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench_off.json 2> $out/bench_off.err &&
timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench_on2.json 2> $out/bench_on2.err && HYPRE_AMD_SMALL_TAIL=0 timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench_off2.json 2> $out/bench_off2.err
python - <<'PY'
import json
for f in ("on", "off", "on2", "off2"):
    for l in open("gpurun_out/r04_step11/bench_%s.json" % f):
        if l.startswith('{"metric"'):
            d = json.loads(l)
            print(f, d["ms_per_step"], d.get("ms_per_step_codes_off"), d["config"].get("coarse_tail_graph_nodes"), d["pcg"]["iterations"], d["pcg"]["ms_per_iteration"])
PY
