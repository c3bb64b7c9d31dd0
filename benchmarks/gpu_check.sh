mkdir -p gpurun_out
OTT_FUZZ_SEEDS=300 python -m pytest tests/test_gpu_vecstore.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_meta.py tests/test_gpu_dist_single.py -x -q > gpurun_out/t_merge.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" gpurun_out/t_merge.log | tail -1
python benchmarks/c1_latency.py 2>&1 | tail -2
python benchmarks/latency_sweep.py 2>&1 | tail -8
