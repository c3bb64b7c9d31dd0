mkdir -p gpurun_out
python benchmarks/nq_sweep.py > gpurun_out/nq_sweep.log 2>&1; grep "^|" gpurun_out/nq_sweep.log
