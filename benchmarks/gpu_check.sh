# scratch script of the last GPU validation run: the driver's three round-end steps + the launcher contract
mkdir -p gpurun_out
OTT_REQUIRE_GPU=1 python -m pytest tests -x -q -m gpu > gpurun_out/t_gpu.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed|^E " gpurun_out/t_gpu.log | head
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc=$? stdout lines=$(wc -l < gpurun_out/bench_final.json)"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_final.json').read())
print({k:d[k] for k in ('value','ms_per_step','n_gpus','parity_checked')}, d['roofline']['frac'], d['roofline']['kernel_ms'], d['extras'].get('config2_score_phase_ms'), d['extras'].get('config2_256q_top100_ms_per_batch'), d['cpu_baseline']['value'])
PY
# launcher contract: --gpus 2 on a 1-GPU box must exit non-zero without a line; torchrun-style 2 ranks over the host transport must print one line
python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/b2.out 2> gpurun_out/b2.err; echo "--gpus 2 on one GPU: rc=$? lines=$(wc -l < gpurun_out/b2.out)"
OTT_BENCH_SINGLE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --rows 2000000 --no-cpu-baseline > gpurun_out/b2t.out 2> gpurun_out/b2t.err; echo "torchrun 2 ranks (host transport): rc=$? lines=$(wc -l < gpurun_out/b2t.out)"; cut -c1-260 gpurun_out/b2t.out
