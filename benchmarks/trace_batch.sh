#!/bin/bash
# Kernel timeline of one batch (default 256 queries, top-100, 10M x 768): every dispatch of the last of three batches, in order.
# bash benchmarks/trace_batch.sh <nq> <outdir>
R=$(cd "$(dirname "$0")/.." && pwd)
NQ=${1:-256}
O=${2:-$R/gpurun_out/trace_$NQ}
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
cd "$R"
rocprofv3 --kernel-trace --output-format csv -d "$O/kt" -- python3 benchmarks/mfma_batch.py $NQ > "$O/run.log" 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
rows = []
for f in glob.glob(O + "/kt/*/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last batch = everything after the third-from-last launch_split/hi_rows kernel
starts = [i for i, r in enumerate(rows) if "hi_rows_kernel" in r["Kernel_Name"] or "split_rows_kernel" in r["Kernel_Name"]]
b = starts[-1] if starts else 0
t0 = int(rows[b]["Start_Timestamp"])
prev_end = t0
for r in rows[b:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-60:]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:6.1f} us  grid {r.get('Grid_Size', '?'):>8}  {name}")
    prev_end = e
print("total %.3f ms" % ((prev_end - t0) / 1e6))
PY
