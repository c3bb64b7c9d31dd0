mkdir -p gpurun_out
OTT_FUZZ_SEEDS=300 python -m pytest tests/test_gpu_mfma.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_bf3_stress.py -x -q -n 4 > gpurun_out/t_m.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" gpurun_out/t_m.log | tail -1
python benchmarks/run_configs.py c2 2>&1 | grep "^| C2"
