#!/bin/bash
# every experiment build under otters_amd/csrc/variants/ on config 2's first level (see i8_tile.py); output: gpurun_out/variants/table.txt
# usage: i8_tile_variants.sh name[:force_fallback] ...
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/variants
rm -rf "$O"; mkdir -p "$O"
cd "$R"
exec < /dev/null
for rep in 1 2; do
for spec in "$@"; do
  lib=${spec%%:*}; ff=0; [ "$spec" != "$lib" ] && ff=${spec##*:}
  OTT_FORCE_FALLBACK=$ff OTT_LIB_PATH=$R/otters_amd/csrc/variants/lib_$lib.so timeout 300 python3 benchmarks/i8_tile.py ${ROWS:-10000000} ${NQ:-256} ${FMT:--1} > "$O/$lib.$ff.$rep.log" 2>&1
  grep -h "^RESULT" "$O/$lib.$ff.$rep.log" | sed "s/^RESULT/RESULT ff=$ff/" | tee -a "$O/table.txt"
  grep -h "ott mfma dbg\] per tile\|^ABL" "$O/$lib.$ff.$rep.log" | tee -a "$O/table.txt"
  grep -h "Error\|error\|Traceback" "$O/$lib.$ff.$rep.log" | head -3
done
done
