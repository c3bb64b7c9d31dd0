#!/bin/bash
# Round-4 evidence behind profiles/round4/: run on the GPU box (bash benchmarks/profile_round4.sh); everything lands in
# gpurun_out/prof4/ and the summaries are then copied into profiles/round4/.  ONE box for all of it.
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/prof4
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
cd "$R"
exec < /dev/null
# 1. the driver's command: plain (its line measures FETCH_SIZE / WRITE_SIZE itself, in PMC child runs), then under kernel trace + stats
timeout 600 python3 bench.py --steps 20 --warmup 5 > "$O/bench_plain.json" 2> "$O/bench_plain.err"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic off > "$O/bench_stats.json" 2> "$O/bench_stats.err"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --traffic off > "$O/bench_fetch.json" 2> "$O/bench_fetch.err"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --traffic off > "$O/bench_write.json" 2> "$O/bench_write.err"
# 2. config-2 batch (256 queries) on the half hi plane: matrix-pipe busy, clock, fetch
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$O/c2_mfma" -- python3 benchmarks/mfma_batch.py 256 > "$O/c2_mfma.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/c2_fetch" -- python3 benchmarks/mfma_batch.py 256 > "$O/c2_fetch.log" 2>&1
# 3. the in-process multi-GPU line (every shard on GPU 0: functional evidence), tables
OTT_BENCH_SINGLE_DEVICE=1 timeout 600 python3 bench.py --gpus 8 --inprocess --rows 1250000 --steps 10 --warmup 2 > "$O/bench_inprocess_8_one_gpu.json" 2> "$O/bench_inprocess.err"
timeout 600 python3 benchmarks/run_configs.py > "$O/run_configs.log" 2>&1
timeout 300 python3 benchmarks/nq_sweep.py > "$O/nq_sweep.md" 2>&1
timeout 900 python3 benchmarks/cliff_hunt.py > "$O/cliff_hunt.md" 2>&1
find "$O" -name "*.csv" | wc -l
tail -2 "$O"/*.json | cut -c1-600
