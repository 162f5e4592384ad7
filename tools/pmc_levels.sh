# rocprofv3 passes over y = A_l x of levels 0..2 of the 256^3 hierarchy (tools/bench_levels_spmv.py); run on the GPU box:
#   bash tools/pmc_levels.sh <tag> [variant]   -> gpurun_out/<tag>/{trace,pmc_*}; then python tools/pmc_levels_summary.py <tag>
# Counters in passes of their own, with no tracing option beside --pmc (the pool refuses the combination); the kernel
# durations come from a separate --kernel-trace --stats run of the same command.
set -e
tag=${1:-r02_levels}
variant=${2:-0}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 tools/bench_levels_spmv.py 256 3 10 $variant > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
echo "done trace"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set -d $out/pmc_$i -o p --output-format csv -- python3 tools/bench_levels_spmv.py 256 3 10 $variant > $out/pmc_$i.log 2>&1 || { echo "set $i ($set) FAILED"; tail -3 $out/pmc_$i.log; continue; }
  echo "done set $i"
done
