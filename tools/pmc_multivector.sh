# rocprofv3 passes over tools/bench_multivector.py (256^3 7-point operator, 1 / 2 / 3 / 4 / 8 columns, codes on and off);
# run on the GPU box:  bash tools/pmc_multivector.sh <tag>   -> gpurun_out/<tag>/{trace,pmc_*};
# then python tools/pmc_multivector_summary.py <tag>.  Counters in passes of their own (no tracing option beside --pmc).
set -e
tag=${1:-r04_mv_pmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 tools/bench_multivector.py 256 10 > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
echo "done trace"
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set -d $out/pmc_$i -o p --output-format csv -- python3 tools/bench_multivector.py 256 3 > $out/pmc_$i.log 2>&1 || { echo "set $i ($set) FAILED"; tail -3 $out/pmc_$i.log; continue; }
  echo "done set $i"
done
