#!/bin/bash
# Round 6: the L2 -> LDS fill path in isolation (benchmarks/cpp/fill_path.hip): the table, then FETCH_SIZE and the L2 hit / miss
# counters per configuration, each --pmc set in its own run.  Run on the GPU box: bash benchmarks/profile_fill_path.sh <outdir>
R=$(cd "$(dirname "$0")/.." && pwd)
O=${1:-$R/gpurun_out/fill_path}
case "$O" in /*) ;; *) O="$R/$O" ;; esac  # (the script changes directory below: the output directory must be absolute)
rm -rf "$O"; mkdir -p "$O"
B="$R/benchmarks/cpp/fill_path"
[ -x "$B" ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 "$R/benchmarks/cpp/fill_path.hip" -o "$B"
cd /tmp && export TMPDIR=/tmp
"$B" > "$O/table.txt" 2>&1
"$B" > "$O/table_second_run.txt" 2>&1
for CFG in rows_hbm rows_hbm_nt queries_l2 rows_l2 both both_qfirst both_l2 both_mfma; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch_$CFG" -- "$B" $CFG > "$O/fetch_$CFG.log" 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$O/tcc_$CFG" -- "$B" $CFG > "$O/tcc_$CFG.log" 2>&1
done
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$O/clk_both_mfma" -- "$B" both_mfma > "$O/clk_both_mfma.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$O/clk_mfma_only" -- "$B" mfma_only > "$O/clk_mfma_only.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d "$O/clk_both" -- "$B" both > "$O/clk_both.log" 2>&1
python3 - "$O" <<'PY' > "$O/summary.txt"
import csv, collections, glob, sys
O = sys.argv[1]
print(open(O + "/table.txt").read())
for d in sorted(glob.glob(O + "/*_*/")):
    tag = d.rstrip("/").split("/")[-1]
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "fill_kernel" not in r["Kernel_Name"]: continue
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if not per: continue
    ids = sorted(per, key=int)[2:]  # past the two warm-up launches
    n = len(ids)
    tot = collections.defaultdict(float)
    for i in ids:
        for k, v in per[i].items(): tot[k] += v / n
    ms = sum(dur[i] for i in ids) / n / 1e6
    line = "%-22s %d dispatches, %.3f ms under the counters:" % (tag, n, ms)
    if "FETCH_SIZE" in tot: line += "  FETCH_SIZE %.0f KiB = %.3f GB per launch (x 1024 x 2, gfx950)" % (tot["FETCH_SIZE"], tot["FETCH_SIZE"] * 2048 / 1e9)
    if "TCC_REQ_sum" in tot: line += "  TCC req %.3g hit %.3g miss %.3g (hit rate %.3f)" % (tot["TCC_REQ_sum"], tot["TCC_HIT_sum"], tot["TCC_MISS_sum"], tot["TCC_HIT_sum"] / max(tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"], 1))
    if "GRBM_GUI_ACTIVE" in tot:
        cyc = tot["GRBM_GUI_ACTIVE"] / 8
        line += "  shader clock %.2f GHz" % (cyc / (ms * 1e6))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in tot: line += ", MFMA busy %.1f %% of SIMD cycles" % (100 * tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4))
    print(line)
PY
cat "$O/summary.txt"
