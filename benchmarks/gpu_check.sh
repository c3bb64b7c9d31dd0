mkdir -p gpurun_out
python -m pytest tests/test_gpu_mfma.py tests/test_gpu_bf3_stress.py tests/test_gpu_fuzz.py -x -q > gpurun_out/t_mfma.log 2>&1; echo "mfma tests rc=$?"; tail -3 gpurun_out/t_mfma.log
python benchmarks/run_configs.py c4 2>&1 | grep "^| C4"
python benchmarks/nq_sweep.py 10000000 768 100 256,512,768,1024 2>&1 | tail -5
