cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r3g; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof_mc -o b --output-format csv -- python3 bench.py --relax 21 --relax-up 22 --no-cpu-baseline > $out/prof_mc.log 2>&1
tail -1 $out/prof_mc.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['setup_seconds'], d['config']['coarse_tail_graph_nodes'], d['vcycle'])"
head -12 $out/prof_mc/b_kernel_stats.csv
