set -x
out=gpurun_out/r04_step21
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5; do
  timeout -k 10 300 python bench.py --no-cpu-baseline > $out/base_$i.json 2> /dev/null
  HYPRE_AMD_LIB=hypre_amd/lib/libhypre_amd_fg4.so timeout -k 10 300 python bench.py --no-cpu-baseline > $out/fg4_$i.json 2> /dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_step21/*.json")):
    for l in open(f):
        if l.startswith('{"metric"'):
            d=json.loads(l); print(f.split('/')[-1], round(d["ms_per_step"],4), round(d["ms_per_step_codes_off"],4), round(d["spmv_level1"]["ms_per_launch"],4))
PY
