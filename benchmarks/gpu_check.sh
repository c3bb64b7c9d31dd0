mkdir -p gpurun_out
timeout 300 python benchmarks/hi256_ab.py 10000000 768 256 100 8 2>&1 | tail -7
