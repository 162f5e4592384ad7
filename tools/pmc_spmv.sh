# Counter passes over the fine-level SpMV (tools/bench_spmv.py), one rocprofv3 run per counter set, on the GPU box:
#   bash tools/pmc_spmv.sh <tag>      -> gpurun_out/<tag>_<set>/...; then  python tools/pmc_summary.py <tag> gpurun_out/<tag>_
set -e
tag=${1:-pmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $set | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --pmc $set -d gpurun_out/${tag}_$name -o p --output-format csv -- python3 tools/bench_spmv.py 256 10 > gpurun_out/${tag}_$name.log 2>&1 || { tail -5 gpurun_out/${tag}_$name.log; exit 1; }
  echo "done $set"
done
