set -x
out=gpurun_out/r04_step17
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof -o b --output-format csv -- python3 bench.py --relax 21 --relax-up 22 --no-cpu-baseline > $out/prof.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r04_step17/prof/**/*kernel_trace.csv', recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if 'mc_small_sweep' in r['Kernel_Name']]
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000 for r in rows]
    print(len(d), rows[0]['Kernel_Name'][:60])
    m=len(d)//2//8*8
    print([round(x,1) for x in d[m:m+16]])
PY
