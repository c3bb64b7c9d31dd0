mkdir -p gpurun_out
OTT_FUZZ_SEEDS=200 python -m pytest tests/test_gpu_mfma.py tests/test_gpu_bf3_stress.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q > gpurun_out/t_mfma.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" gpurun_out/t_mfma.log | tail -1
python benchmarks/hi256_ab.py 10000000 768 256 100 8 2>&1 | tail -6
python benchmarks/nq_sweep.py 10000000 768 100 64,128,256,1024 2>&1 | tail -4
