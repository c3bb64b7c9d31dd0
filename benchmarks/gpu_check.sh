mkdir -p gpurun_out
nproc; free -g | head -2
python -m pytest tests -m gpu -x -q --durations=30 > gpurun_out/t_gpu_serial.log 2>&1; echo "gpu tests rc=$?"; tail -45 gpurun_out/t_gpu_serial.log
