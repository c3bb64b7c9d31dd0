#!/bin/bash
# Round-2 evidence behind profiles/round2/: run on the GPU box (bash benchmarks/profile_round2.sh); everything lands in
# gpurun_out/prof2/ and the summaries are then copied into profiles/round2/ by hand.
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/prof2
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
cd "$R"
# 1. the driver's command under kernel trace + stats, then FETCH_SIZE / WRITE_SIZE in their own passes
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$O/bench_stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > "$O/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > "$O/bench_write.log" 2>&1
python3 bench.py --steps 50 --warmup 5 > "$O/bench_plain.log" 2>&1
# 2. the sharded path at one rank (RCCL communicator of one rank): kernel trace shows what the exchange adds
OTT_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/dist1_stats" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > "$O/bench_dist1_stats.log" 2>&1
OTT_BENCH_FORCE_DIST=1 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > "$O/bench_dist1_plain.log" 2>&1
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > "$O/bench_nodist_plain.log" 2>&1
# 3. config-2 batch (256 queries): PMC triple of the hi pass, both kernels
for HI in 0 1; do
  OTT_HI256=$HI rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$O/c2_hi${HI}_mfma" -- python3 benchmarks/mfma_batch.py 256 > "$O/c2_hi${HI}_mfma.log" 2>&1
  OTT_HI256=$HI rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d "$O/c2_hi${HI}_waits" -- python3 benchmarks/mfma_batch.py 256 > "$O/c2_hi${HI}_waits.log" 2>&1
  OTT_HI256=$HI rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/c2_hi${HI}_fetch" -- python3 benchmarks/mfma_batch.py 256 > "$O/c2_hi${HI}_fetch.log" 2>&1
done
# 4. tables
python3 benchmarks/hi256_ab.py 10000000 768 256 100 8 > "$O/hi256_ab.log" 2>&1
python3 benchmarks/c1_latency.py > "$O/c1_latency.log" 2>&1
python3 benchmarks/sharded_batch.py > "$O/sharded_batch.log" 2>&1
python3 benchmarks/run_configs.py > "$O/run_configs.log" 2>&1
find "$O" -name "*.csv" | wc -l
tail -2 "$O"/*.log | cut -c1-600
