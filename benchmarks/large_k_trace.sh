#!/bin/bash
# Kernel timeline of one large-k query: bash benchmarks/large_k_trace.sh <k> <outdir>
R=$(cd "$(dirname "$0")/.." && pwd)
K=${1:-1000}
O=${2:-$R/gpurun_out/trace_k$K}
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
cd "$R"
rocprofv3 --kernel-trace --output-format csv -d "$O/kt" -- python3 benchmarks/large_k_trace.py $K > "$O/run.log" 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
rows = []
for f in glob.glob(O + "/kt/*/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last query = from the third-from-last exact_kernel dispatch of the smallest grid pattern: take the last third of the dispatches after the generator
gen = max(i for i, r in enumerate(rows) if "rand_fill" in r["Kernel_Name"] or "inv_norm" in r["Kernel_Name"] or "min_pos" in r["Kernel_Name"])
q = rows[gen + 1:]
per = len(q) // 3
q = q[-per:]
t0 = int(q[0]["Start_Timestamp"]); prev = t0
for r in q:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {(s - prev) / 1e3:6.1f} us  {r['Kernel_Name'].split('(')[0][-70:]}")
    prev = e
print("total %.3f ms" % ((prev - t0) / 1e6))
print(open(O + "/run.log").read().strip().splitlines()[-1])
PY
