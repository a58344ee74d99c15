set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_transformer.py -x -q -k "feed_forward" --timeout 200 --timeout-method thread > gpurun_out/r3f_t.log 2>&1 || { tail -25 gpurun_out/r3f_t.log; exit 1; }
tail -2 gpurun_out/r3f_t.log
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_ffn
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_ffn -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/ubench/ffn_time.py > $GRAFT_REPO_ROOT/gpurun_out/r3f_ffn.txt 2>&1 || true
grep tokens $GRAFT_REPO_ROOT/gpurun_out/r3f_ffn.txt
head -5 $GRAFT_REPO_ROOT/gpurun_out/prof_ffn/p_kernel_stats.csv | cut -c1-160
