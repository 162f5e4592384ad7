#!/bin/bash
# copy the judged summaries of tools/r03_measure.sh (+ tools/pmc_levels.sh) from gpurun_out/ (scratch) into profiles/ (tracked)
set -e
m=gpurun_out/${1:-r03_measure}
for f in c2 c4 c4_again c5 mc128 mc256 rehearsal_dev2 rehearsal_host2 rehearsal_dev4; do tail -1 $m/bench_$f.json > profiles/r03_bench_$f.json; done
cp $m/prof_c2/b_kernel_stats.csv profiles/r03_bench256_kernel_stats.csv
cp $m/prof_c4/b_kernel_stats.csv profiles/r03_c4_27pt256_tsgs_kernel_stats.csv
cp $m/prof_c5/b_kernel_stats.csv profiles/r03_c5_difconv256_mixed_kernel_stats.csv
python tools/trace_summary.py $m/prof_c2/b_kernel_trace.csv profiles/r03_bench256_roofline_kernel_from_trace.json
python - <<PY
import json
out = {"7pt": json.load(open("$m/levels_7pt.json")), "27pt_relax11": json.load(open("$m/levels_27pt.json"))}
json.dump(out, open("profiles/r03_levels_ops.json", "w"), indent=1)
PY
grep -h "setup [0-9] \|matrix generation\|setup level\|product:\|interpolation:" $m/setup_device.log $m/bench_rehearsal_dev2.err $m/bench_rehearsal_host2.err > profiles/r03_setup_timing.txt || true
