set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 120 python tools/bench_kernels.py --mode 1 --iters 10 --only wgrad_pair
rm -rf $O/pmc_pf $O/pmc_pw
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_pf -- python3 tools/bench_kernels.py --mode 1 --only wgrad_pair --iters 3 > $O/pmc_pf.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_pw -- python3 tools/bench_kernels.py --mode 1 --only wgrad_pair --iters 3 > $O/pmc_pw.log 2>&1
echo pmc done
