mkdir -p gpurun_out
python -m pytest tests/test_gpu_dist_single.py tests/test_gpu_vecstore.py tests/test_gpu_cpp_mirror.py -x -q > gpurun_out/t_dist.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" gpurun_out/t_dist.log | tail -1
for i in 1 2; do
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain  ', d['ms_per_step'], d['roofline']['kernel_ms'])"
OTT_BENCH_FORCE_DIST=1 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sharded', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
python benchmarks/sharded_batch.py 2>/dev/null | grep "^|"
