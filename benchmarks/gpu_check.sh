mkdir -p gpurun_out
python -m pytest tests/test_gpu_mfma.py -x -q -n 4 > gpurun_out/t_m.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/t_m.log
