#!/bin/bash
# round 4, step 8: the whole GPU suite with the fused multivector products, their timing, a kernel trace of it
set -x
out=gpurun_out/r04_step8
mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=5 > $out/tests.log 2>&1
rc=$?
tail -12 $out/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/bench_multivector.py 256 20 --json $out/mv_7pt.json > $out/mv_7pt.log 2>&1 && grep -c fused $out/mv_7pt.log &&
timeout -k 10 400 python tools/bench_multivector.py 160 20 --stencil 27 --json $out/mv_27pt.json > $out/mv_27pt.log 2>&1 &&
(cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $out/prof -o mv --output-format csv -- python3 tools/bench_multivector.py 256 10 > $out/prof.log 2>&1) &&
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r04_step8/prof/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        if 'mv_kernel' in r['Name'] or 'spmv_sl_kernel<0' in r['Name'] or 'spmv_xs_kernel<0, 0' in r['Name']:
            print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
