#!/bin/bash
# copy the judged summaries of tools/r02_measure.sh + tools/pmc_levels.sh (gpurun_out/, scratch) into profiles/ (tracked)
set -e
m=gpurun_out/r02_measure
for f in c2 c4 c5 mc128 mc256 gs128 2rank_rehearsal; do tail -1 $m/bench_$f.json > profiles/r02_bench_$f.json; done
cp $m/prof_c2/b_kernel_stats.csv profiles/r02_bench256_kernel_stats.csv
cp $m/prof_c4/b_kernel_stats.csv profiles/r02_c4_27pt256_tsgs_kernel_stats.csv
cp $m/prof_c5/b_kernel_stats.csv profiles/r02_c5_difconv256_mixed_kernel_stats.csv
python tools/trace_summary.py $m/prof_c2/b_kernel_trace.csv profiles/r02_bench256_roofline_kernel_from_trace.json
l=gpurun_out/r02_levels_xs
cp $l/trace/t_kernel_stats.csv profiles/r02_levels_xs_kernel_stats.csv
cp $l/summary.json profiles/r02_levels_xs_summary.json 2>/dev/null || true
cp $m/prof_setup/s_kernel_stats.csv profiles/r02_setup256_kernel_stats.csv
grep -h "setup [0-9] \|matrix generation\|setup level\|product:\|interpolation:" $m/setup_device.log $m/setup_host.log > profiles/r02_setup256_timing.txt
