# round 4, step 20: suite (graph on through tests/conftest.py), bench lines with the default (eager) cycle
set -x
out=gpurun_out/r04_step20
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1
rc=$?
tail -4 $out/tests.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 300 python bench.py > $out/bench_c2.json 2> $out/bench_c2.err; echo "c2 exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof_c2 -o b --output-format csv -- python3 bench.py --no-cpu-baseline > $out/prof_c2.log 2>&1; echo "prof exit $?"
timeout -k 10 400 python bench.py --problem 27pt --relax 11 > $out/bench_c4.json 2> $out/bench_c4.err; echo "c4 exit $?"
timeout -k 10 300 python bench.py --problem difconv --mixed > $out/bench_c5.json 2> $out/bench_c5.err; echo "c5 exit $?"
timeout -k 10 300 python bench.py --relax 21 --relax-up 22 --cpu-cycles 1 > $out/bench_mc256.json 2> $out/bench_mc256.err; echo "mc256 exit $?"
python - <<'PY'
import json
for f in ("c2", "c4", "c5", "mc256"):
    for l in open("gpurun_out/r04_step20/bench_%s.json" % f):
        if l.startswith('{"metric"'):
            d = json.loads(l)
            print(f, round(d["ms_per_step"],4), round(d["value"]/1e9,3), round(d["roofline"]["frac"],3), round(d.get("ms_per_step_codes_off") or 0,4), d["config"].get("coarse_tail_graph_nodes"), d["config"].get("one_workgroup_tail_from_level"), d["pcg"]["iterations"], round(d["pcg"]["solve_ms"],2), d["cpu_baseline"]["gpu_vs_cpu_cycle_rel_max_diff"])
PY
