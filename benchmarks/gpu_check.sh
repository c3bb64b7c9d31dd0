mkdir -p gpurun_out
python -m pytest tests/test_gpu_mfma.py tests/test_gpu_meta.py tests/test_gpu_fullsize.py tests/test_gpu_dist_single.py -x -q > gpurun_out/t_m.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/t_m.log
python benchmarks/run_configs.py c3 2>&1 | grep "^| C3"
