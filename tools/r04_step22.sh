# round 4, step 22: kernel statistics of the C4, C5 and multicolour bench commands
set -x
out=gpurun_out/r04_step22
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/prof_c4 -o b --output-format csv -- python3 bench.py --problem 27pt --relax 11 --no-cpu-baseline > $out/prof_c4.log 2>&1; echo "c4 exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof_c5 -o b --output-format csv -- python3 bench.py --problem difconv --mixed --no-cpu-baseline > $out/prof_c5.log 2>&1; echo "c5 exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof_mc -o b --output-format csv -- python3 bench.py --relax 21 --relax-up 22 --no-cpu-baseline > $out/prof_mc.log 2>&1; echo "mc exit $?"
ls $out/prof_c4 $out/prof_c5 $out/prof_mc
