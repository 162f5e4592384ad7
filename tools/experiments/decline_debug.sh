out=gpurun_out/r3h; mkdir -p $out
export HYPRE_AMD_DEBUG_FREE=1 OMP_NUM_THREADS=1
spec='{"batch": [{"name": "d", "options": {"n": [32, 32, 16], "P": [2, 2, 1], "relax_type": 18, "coarsen_type": 8}, "device": 1, "compare_setup": 1, "decline": [1, 1]}], "transport": "staged"}'
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29544 tests/dist_worker.py "$spec" > $out/decline.out 2> $out/decline.err
echo rc=$?
grep -n "libhypre_amd\|RESULT" $out/decline.err $out/decline.out | head -40
