#!/bin/bash
# i8_tile.py for several batch sizes and builds: NQS="64 128" bash benchmarks/i8_tile_nq.sh name ...   (output: gpurun_out/variants/nq_table.txt)
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/variants
mkdir -p "$O"; rm -f "$O/nq_table.txt"
cd "$R"
exec < /dev/null
for rep in 1 2; do
for nq in ${NQS:-64 128}; do
for lib in "$@"; do
  OTT_LIB_PATH=$R/otters_amd/csrc/variants/lib_$lib.so timeout 300 python3 benchmarks/i8_tile.py ${ROWS:-10000000} $nq ${FMT:--1} 2>&1 | grep -h "^RESULT\|rror" | tee -a "$O/nq_table.txt"
done
done
done
