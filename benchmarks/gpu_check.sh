mkdir -p gpurun_out
OTT_FUZZ_SEEDS=2500 timeout 1700 python -m pytest tests/test_gpu_fuzz.py -x -q > gpurun_out/soak_default.log 2>&1; echo "soak default rc=$?"; tail -2 gpurun_out/soak_default.log
