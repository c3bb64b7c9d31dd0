#!/bin/bash
# Round-6 evidence behind profiles/round6/: run on the GPU box (bash benchmarks/profile_round6.sh); everything lands in
# gpurun_out/prof6/ and the summaries are then copied into profiles/round6/.  ONE box for all of it.
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/prof6
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
cd "$R"
exec < /dev/null
# 1. the driver's command: plain (its line measures FETCH_SIZE / WRITE_SIZE itself, in PMC child runs), then under kernel trace + stats
timeout 600 python3 bench.py --steps 20 --warmup 5 > "$O/bench_plain.json" 2> "$O/bench_plain.err"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic off > "$O/bench_stats.json" 2> "$O/bench_stats.err"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --traffic off > "$O/bench_fetch.json" 2> "$O/bench_fetch.err"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --traffic off > "$O/bench_write.json" 2> "$O/bench_write.err"
# 2. the int8 candidate pass (config 2's shape and a single query) next to the half hi pass: matrix-pipe busy, clock, fetch, kernel times
bash benchmarks/profile_i8.sh "$O/i8" > "$O/i8_summary.txt" 2>&1
# 3. the single-query int8 sweep under the kernel trace (AUTO on the headline store) and its fetch
cat > "$O/single_auto.py" <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from otters_amd import Metric, Path, VecStore
s = VecStore(768); s.reserve(10_000_000); s.append_random(10_000_000, 0x07735)
t0 = time.perf_counter()
while not s.batch_ready() and time.perf_counter() - t0 < 10: time.sleep(0.01)
q = np.random.default_rng(1).uniform(-1, 1, (30, 768)).astype(np.float32)
for i in range(30):
    t = time.perf_counter(); s.query(q[i], Metric.Cosine).take(10).collect_arrays(); dt = time.perf_counter() - t
st = s.last_stats; print("single query via AUTO: wall %.3f ms, score %.3f ms, merge %.3f ms, path %d" % (dt * 1e3, st["score_ns"] / 1e6, st["merge_ns"] / 1e6, st["path_used"]))
PY
timeout 300 python3 "$O/single_auto.py" > "$O/single_auto_plain.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/single_stats" -- python3 "$O/single_auto.py" > "$O/single_auto_stats.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/single_fetch" -- python3 "$O/single_auto.py" > "$O/single_auto_fetch.log" 2>&1
# 4. tables
timeout 600 python3 benchmarks/run_configs.py > "$O/run_configs.log" 2>&1
timeout 300 python3 benchmarks/i8_level.py > "$O/i8_level.md" 2>&1
timeout 600 python3 benchmarks/auto_choice.py 768 150000,300000,500000,1000000,3000000,10000000 > "$O/auto_choice.md" 2>&1
timeout 300 python3 benchmarks/auto_choice.py 128 1000000,4000000 >> "$O/auto_choice.md" 2>&1
OTT_BENCH_SINGLE_DEVICE=1 timeout 600 python3 bench.py --gpus 8 --inprocess --rows 1250000 --steps 10 --warmup 2 > "$O/bench_inprocess_8_one_gpu.json" 2> "$O/bench_inprocess.err"
# the N > 1 line with its extras (config 4's shape, the strong split), 8 ranks on this ONE GPU over RCCL's socket transport: functional evidence
OTT_BENCH_SINGLE_DEVICE=rccl timeout 900 python3 bench.py --gpus 8 --rows 1250000 --steps 10 --warmup 2 > "$O/bench_8_ranks_one_gpu_rccl.json" 2> "$O/bench_8_ranks.err"
OTT_BENCH_SINGLE_DEVICE=rccl timeout 900 python3 bench.py --gpus 2 --rows 10000000 --steps 10 --warmup 2 > "$O/bench_2_ranks_one_gpu_rccl_10M.json" 2> "$O/bench_2_ranks.err"
timeout 900 python3 benchmarks/clustered_10m.py > "$O/clustered_10m.md" 2>&1
find "$O" -name "*.csv" | wc -l
tail -2 "$O"/*.json | cut -c1-700
cat "$O/single_auto_plain.log" "$O/i8_summary.txt" | head -40
