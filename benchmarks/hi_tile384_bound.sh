#!/bin/bash
# kernel time of the hi pass's candidate kernel per 256-query batch, today's tile against the fill of a 384 x 256 tile (see hi_tile384_bound.py)
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/tile384
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
cd "$R"
exec < /dev/null
export OTT_LIB_PATH=$R/otters_amd/csrc/libotters_hip_dbg.so
NQ=${1:-256}   # queries per batch: 256 (kernel <4, true, 4>), 128 (<2, true, 4>), 64 (<1, true, 4>)
KN=$([ "$NQ" -gt 128 ] && echo 4 || ([ "$NQ" -gt 64 ] && echo 2 || echo 1))
for abl in 0 16 32 0; do
  d="$O/abl${abl}_$RANDOM"
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$d" -o t -- python3 benchmarks/hi_tile384_bound.py $abl 10000000 $NQ > "$d.log" 2>&1
  f=$(find "$d" -name "*kernel_trace.csv" | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" $abl $KN <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if f"mfma_score_kernel<{sys.argv[3]}, true, 4>" in r["Kernel_Name"]]  # the hi pass's candidate kernel only
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
# a batch's candidate pass = its dispatches up to the cascade's next level: group by the 4 batches (equal counts when nothing falls through)
big = sorted(d, reverse=True)[:4]  # the last (largest) round of each of the 4 batches
print(f"| abl {sys.argv[2]} | dispatches {len(d)} | total {sum(d):.2f} ms | four largest dispatches (the last round of each batch): " + ", ".join(f"{x:.3f}" for x in big) + " ms |")
PY
  fi
  grep "^abl" "$d.log"
done
