mkdir -p gpurun_out
python -m pytest tests/test_gpu_dist_single.py -x -q --durations=5 > gpurun_out/t_dist.log 2>&1; echo "dist tests rc=$?"; grep -E "passed|failed|s call|s setup" gpurun_out/t_dist.log | tail -8
