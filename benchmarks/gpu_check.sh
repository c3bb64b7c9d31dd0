mkdir -p gpurun_out
for nq in 16 256; do echo "== nq $nq"; timeout 600 bash benchmarks/trace_batch.sh $nq 2>&1 | grep -E "select|total"; done
python -m pytest tests/test_gpu_mfma.py tests/test_gpu_fuzz.py tests/test_gpu_bf3_stress.py tests/test_gpu_fullsize.py -x -q > gpurun_out/t_m.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^E " gpurun_out/t_m.log | head
