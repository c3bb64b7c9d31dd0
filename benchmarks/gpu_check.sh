mkdir -p gpurun_out
OTT_FUZZ_SEEDS=200 python -m pytest tests/test_gpu_vecstore.py tests/test_gpu_fuzz.py tests/test_gpu_dist_single.py tests/test_gpu_meta.py -x -q > gpurun_out/t_sort.log 2>&1; echo "large-k related tests rc=$?"; tail -3 gpurun_out/t_sort.log | cut -c1-300
python benchmarks/k_sweep.py 2>&1 | tail -12
