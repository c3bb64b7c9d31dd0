# last GPU validation of the round: the GPU test suite, smoke(), the driver's bench command, the launcher contract
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/t_all.log 2>&1; echo "ALL gpu tests rc=$?"; grep -E "passed|failed" gpurun_out/t_all.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-160
python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$? lines on stdout: $(wc -l < gpurun_out/bench.json)"; cut -c1-330 gpurun_out/bench.json
OTT_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/bench_d.json 2> gpurun_out/bench_d.err; echo "sharded (1-rank RCCL) rc=$? lines on stdout: $(wc -l < gpurun_out/bench_d.json)"
OTT_BENCH_SINGLE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --rows 2000000 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/bench_tr.json 2> gpurun_out/bench_tr.err; echo "torchrun 2 ranks (host transport, one GPU) rc=$? lines on stdout: $(wc -l < gpurun_out/bench_tr.json)"
timeout 120 python bench.py --gpus 2 --rows 100000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/bench_fail.json 2> gpurun_out/bench_fail.err; echo "--gpus 2 on a 1-GPU box rc=$? (expected 1), stdout lines: $(wc -l < gpurun_out/bench_fail.json)"
