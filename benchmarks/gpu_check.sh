mkdir -p gpurun_out
OTT_FUZZ_SEEDS=200 python -m pytest tests/test_gpu_vecstore.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_meta.py -x -q > gpurun_out/t_merge.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/t_merge.log | cut -c1-200
python benchmarks/c1_latency.py 2>&1 | tail -2
