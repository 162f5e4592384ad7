# AMG-PCG on the C1 / C2 problem class with the smoothers the GPU offers: iterations to 1e-8, time per cycle, time to solution
out=gpurun_out/${1:-r03_smoothers}
mkdir -p $out
for n in 64 128; do
  for sm in "18 -1 l1jacobi" "11 -1 twostage1" "12 -1 twostage2" "21 22 multicolour" "13 14 hybrid_l1gs" "3 4 hybrid_gs"; do
    set -- $sm
    extra=""; [ "$2" != "-1" ] && extra="--relax-up $2"
    timeout -k 10 300 python bench.py --grid $n --relax $1 $extra --no-cpu-baseline --steps 10 > $out/s_${n}_$3.json 2> $out/s_${n}_$3.err; echo "$n $3 exit $?"
  done
done
for sm in "18 -1 l1jacobi" "11 -1 twostage1" "21 22 multicolour"; do
  set -- $sm
  extra=""; [ "$2" != "-1" ] && extra="--relax-up $2"
  timeout -k 10 300 python bench.py --grid 256 --relax $1 $extra --no-cpu-baseline --steps 10 > $out/s_256_$3.json 2> $out/s_256_$3.err; echo "256 $3 exit $?"
done
python - <<PY
import glob, json, os
rows = []
for f in sorted(glob.glob("$out/s_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    name = os.path.basename(f)[2:-5]
    n, sm = name.split("_", 1)
    p = d.get("pcg") or {}
    rows.append({"grid": int(n), "smoother": sm, "ms_per_cycle": d["ms_per_step"], "pcg_iterations": p.get("iterations"),
                 "pcg_ms_per_iteration": p.get("ms_per_iteration"), "pcg_solve_ms": p.get("solve_ms"),
                 "final_rel_resid": p.get("final_rel_resid"), "setup_seconds": d["config"]["setup_seconds"]})
json.dump({"what": "AMG-PCG to 1e-8 on n^3 7-point Laplacians (PMIS / ext+i(4), V(1,1)), one MI355X: smoothers compared", "rows": rows},
          open("$out/summary.json", "w"), indent=1)
for r in rows: print(r)
PY
