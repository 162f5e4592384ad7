#!/bin/bash
# copy the judged summaries of tools/r03_measure3.sh (final pass of round 3: value codes + slice form) from gpurun_out/ into
# profiles/; the earlier passes' files (r03_bench_*.json of tools/r03_measure.sh: before the value codes) move to r03a_*
set -e
m=gpurun_out/${1:-r03_measure3}
for f in c2 c4 c5 mc256 rehearsal_dev2 rehearsal_dev4; do
  [ -f profiles/r03_bench_$f.json ] && [ ! -f profiles/r03a_bench_$f.json ] && git mv profiles/r03_bench_$f.json profiles/r03a_bench_$f.json
  tail -1 $m/bench_$f.json > profiles/r03_bench_$f.json
done
for f in c2_noslice c2_nocodes c4_nocodes; do tail -1 $m/bench_$f.json > profiles/r03_bench_$f.json; done
for f in r03_bench256_kernel_stats.csv r03_c4_27pt256_tsgs_kernel_stats.csv r03_c5_difconv256_mixed_kernel_stats.csv r03_levels_ops.json r03_bench256_roofline_kernel_from_trace.json; do
  [ -f profiles/$f ] && [ ! -f profiles/r03a_${f#r03_} ] && git mv profiles/$f profiles/r03a_${f#r03_}
done
cp $m/prof_c2/b_kernel_stats.csv profiles/r03_bench256_kernel_stats.csv
cp $m/prof_c4/b_kernel_stats.csv profiles/r03_c4_27pt256_tsgs_kernel_stats.csv
cp $m/prof_c5/b_kernel_stats.csv profiles/r03_c5_difconv256_mixed_kernel_stats.csv
python tools/trace_summary.py $m/prof_c2/b_kernel_trace.csv profiles/r03_bench256_roofline_kernel_from_trace.json
python - <<PY
import json
out = {"7pt": json.load(open("$m/levels_7pt.json")), "27pt_relax11": json.load(open("$m/levels_27pt.json"))}
json.dump(out, open("profiles/r03_levels_ops.json", "w"), indent=1)
PY
grep -h "setup [0-9] \|matrix generation\|setup level\|product:\|interpolation:" $m/setup_device.log $m/bench_rehearsal_dev2.err > profiles/r03_setup_timing_final.txt || true
