mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/t_gpu.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed|^E " gpurun_out/t_gpu.log | head
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc=$? lines=$(wc -l < gpurun_out/bench_final.json)"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_final.json').read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['extras'].get('config2_256q_top100_ms_per_batch'), d['extras'].get('config2_score_phase_ms'), d['extras'].get('single_query_via_cascade_ms'))
PY
python benchmarks/run_configs.py c2 c4 2>&1 | grep "^| C"
