set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python bench.py --steps 20 --warmup 5 > gpurun_out/bench1.json 2> gpurun_out/bench1.err; echo "bench rc $?"; tail -c 3000 gpurun_out/bench1.json; tail -5 gpurun_out/bench1.err
OTT_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/bench1_dist.json 2> gpurun_out/bench1_dist.err; echo "dist rc $?"; cat gpurun_out/bench1_dist.json; tail -5 gpurun_out/bench1_dist.err
OTT_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --rows 3000000 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/bench1_2r.json 2> gpurun_out/bench1_2r.err; echo "2rank rc $?"; cat gpurun_out/bench1_2r.json; tail -5 gpurun_out/bench1_2r.err
python bench.py --gpus 2 --rows 100000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/bench1_fail.json 2> gpurun_out/bench1_fail.err; echo "2gpu-on-1gpu-box rc $? (expected non-zero)"; tail -3 gpurun_out/bench1_fail.err
