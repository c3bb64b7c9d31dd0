mkdir -p gpurun_out
OTT_FUZZ_SEEDS=1500 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -x -q > gpurun_out/soak_default.log 2>&1; echo "soak default rc=$?"; tail -1 gpurun_out/soak_default.log
OTT_HI256=1 OTT_FUZZ_SEEDS=300 timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q > gpurun_out/soak_hi256.log 2>&1; echo "soak hi256 rc=$?"; tail -1 gpurun_out/soak_hi256.log
OTT_NO_HI_PASS=1 OTT_FUZZ_SEEDS=300 timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q > gpurun_out/soak_split.log 2>&1; echo "soak split rc=$?"; tail -1 gpurun_out/soak_split.log
OTT_MFMA_F32=1 OTT_FUZZ_SEEDS=300 timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q > gpurun_out/soak_f32.log 2>&1; echo "soak f32 rc=$?"; tail -1 gpurun_out/soak_f32.log
