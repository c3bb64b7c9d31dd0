#!/bin/bash
# rocprofv3 passes behind profiles/: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own runs
# (counters are never combined with other trace domains).  Run on the GPU box: bash benchmarks/profile_bench.sh <outdir>
R=$(cd "$(dirname "$0")/.." && pwd)
O=${1:-$R/gpurun_out/prof}
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
cd "$R"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$O/bench_stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > "$O/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > "$O/bench_write.log" 2>&1
python3 bench.py --steps 50 --warmup 5 > "$O/bench_plain.log" 2>&1
find "$O" -name "*.csv" | head -20
tail -1 "$O/bench_plain.log" | cut -c1-700
