#!/bin/bash
# Round-3 evidence behind profiles/round3/: run on the GPU box (bash benchmarks/profile_round3.sh); everything lands in
# gpurun_out/prof3/ and the summaries are then copied into profiles/round3/.
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/prof3
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
cd "$R"
# 1. the driver's command: plain (its line measures FETCH_SIZE / WRITE_SIZE itself, in PMC child runs), then under kernel trace + stats
python3 bench.py --steps 20 --warmup 5 > "$O/bench_plain.json" 2> "$O/bench_plain.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic off > "$O/bench_stats.json" 2> "$O/bench_stats.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --traffic off > "$O/bench_fetch.json" 2> "$O/bench_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --traffic off > "$O/bench_write.json" 2> "$O/bench_write.err"
# 2. config-2 batch (256 queries) on the half hi plane: matrix-pipe busy, clock, fetch
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$O/c2_mfma" -- python3 benchmarks/mfma_batch.py 256 > "$O/c2_mfma.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/c2_fetch" -- python3 benchmarks/mfma_batch.py 256 > "$O/c2_fetch.log" 2>&1
# 3. tables
python3 benchmarks/small_store_latency.py 768,128 > "$O/small_store_latency.md" 2>&1
python3 benchmarks/k_sweep.py > "$O/k_sweep.md" 2>&1
python3 benchmarks/nq_sweep.py > "$O/nq_sweep.md" 2>&1
python3 benchmarks/hi_fmt_ab.py > "$O/hi_fmt_ab.md" 2>&1
python3 benchmarks/run_configs.py > "$O/run_configs.log" 2>&1
python3 benchmarks/c1_latency.py > "$O/c1_latency.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/sortprof" -- python3 benchmarks/sort_probe.py > "$O/sortprof.log" 2>&1
find "$O" -name "*.csv" | wc -l
tail -3 "$O"/*.md "$O"/*.log | cut -c1-400
