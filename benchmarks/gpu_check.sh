mkdir -p gpurun_out
python -m pytest tests/test_gpu_vecstore.py tests/test_gpu_meta.py tests/test_gpu_fuzz.py -x -q > gpurun_out/t_v.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^E " gpurun_out/t_v.log | head
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_inv -- python3 $GRAFT_REPO_ROOT/benchmarks/mfma_batch.py 16 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/prof_inv -name "*kernel_stats.csv" | head -1); grep -E "inv_norm|hi_rows|rand_fill" $f | cut -c1-160
