#!/bin/bash
# round 4, step 7: fused multivector products — tests, then the 256^3 timing
set -x
out=gpurun_out/r04_step7
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_seq_matvec_gpu.py -m gpu -x -q > $out/tests.log 2>&1
rc=$?
tail -5 $out/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/bench_multivector.py 256 20 --json $out/mv_7pt.json > $out/mv_7pt.log 2>&1 && tail -12 $out/mv_7pt.log &&
timeout -k 10 400 python tools/bench_multivector.py 160 20 --stencil 27 --json $out/mv_27pt.json > $out/mv_27pt.log 2>&1 && tail -12 $out/mv_27pt.log &&
timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.json
