set -x
out=gpurun_out/r04_step19
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # name, args...
  name=$1; shift
  for i in 1 2; do
    timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/${name}_g1_$i.json 2> /dev/null
    HYPRE_AMD_CYCLE_GRAPH=0 timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/${name}_g0_$i.json 2> /dev/null
  done
}
run j128 --grid 128
run j64 --grid 64
run c4 --problem 27pt --relax 11
run c5 --problem difconv --mixed
run mc --relax 21 --relax-up 22
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_step19/*_g*.json")):
    for l in open(f):
        if l.startswith('{"metric"'):
            d=json.loads(l); print(f.split('/')[-1], round(d["ms_per_step"],4), d["config"]["coarse_tail_graph_nodes"], round(d["pcg"]["ms_per_iteration"],4))
PY
