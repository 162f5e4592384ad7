# round 4, step 16: look-ahead multicolour sweeps — tests, bench of the multicolour cycle with and without
set -x
out=gpurun_out/r04_step16
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_multicolor_gpu.py -m gpu -x -q > $out/tests.log 2>&1
rc=$?
tail -12 $out/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --relax 21 --relax-up 22 --no-cpu-baseline > $out/mc_on.json 2> $out/mc_on.err && HYPRE_AMD_MC_LOOK_AHEAD=0 timeout -k 10 300 python bench.py --relax 21 --relax-up 22 --no-cpu-baseline > $out/mc_off.json 2> $out/mc_off.err
python - <<'PY'
import json
for f in ("on", "off"):
    for l in open("gpurun_out/r04_step16/mc_%s.json" % f):
        if l.startswith('{"metric"'):
            d = json.loads(l)
            print(f, d["ms_per_step"], d["pcg"]["iterations"], d["pcg"]["ms_per_iteration"])
PY
