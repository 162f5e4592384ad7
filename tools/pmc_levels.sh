# rocprofv3 counter passes over the SpMV of one hierarchy level (tools/bench_level_spmv.py); run on the GPU box:
#   bash tools/pmc_levels.sh <level> ; results under gpurun_out/pmc_lev/
set -e
lev=${1:-1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_lev
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set -d gpurun_out/pmc_lev/L${lev}_$i -o p --output-format csv -- python3 tools/bench_level_spmv.py 256 $lev 10 > gpurun_out/pmc_lev/L${lev}_$i.log 2>&1 || { tail -5 gpurun_out/pmc_lev/L${lev}_$i.log; exit 1; }
  echo "done L$lev set $i"
done
