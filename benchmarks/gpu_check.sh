mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/t_gpu.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed|^E " gpurun_out/t_gpu.log | head
python benchmarks/run_configs.py head c2 c3 2>&1 | grep "^| "
