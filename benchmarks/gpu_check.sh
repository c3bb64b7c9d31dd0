mkdir -p gpurun_out
OTT_FUZZ_SEEDS=300 python -m pytest tests/test_gpu_fuzz.py -x -q > gpurun_out/t_fuzz300.log 2>&1; echo "fuzz soak (300 seeds, default) rc=$?"; tail -2 gpurun_out/t_fuzz300.log | cut -c1-300
OTT_HI256=1 OTT_FUZZ_SEEDS=150 python -m pytest tests/test_gpu_fuzz.py -x -q > gpurun_out/t_fuzz_hi256.log 2>&1; echo "fuzz soak (150 seeds, hi256 kernel) rc=$?"; tail -2 gpurun_out/t_fuzz_hi256.log | cut -c1-300
OTT_NO_HI_PASS=1 OTT_FUZZ_SEEDS=100 python -m pytest tests/test_gpu_fuzz.py -x -q > gpurun_out/t_fuzz_split.log 2>&1; echo "fuzz soak (100 seeds, split pass only) rc=$?"; tail -2 gpurun_out/t_fuzz_split.log | cut -c1-300
OTT_MFMA_F32=1 OTT_FUZZ_SEEDS=100 python -m pytest tests/test_gpu_fuzz.py -x -q > gpurun_out/t_fuzz_f32.log 2>&1; echo "fuzz soak (100 seeds, f32 pipe) rc=$?"; tail -2 gpurun_out/t_fuzz_f32.log | cut -c1-300
