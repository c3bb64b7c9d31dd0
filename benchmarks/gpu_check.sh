mkdir -p gpurun_out
python -m pytest tests/test_gpu_fullsize.py -x -q -s > gpurun_out/t_full.log 2>&1; echo "fullsize tests rc=$?"; tail -12 gpurun_out/t_full.log | cut -c1-400
