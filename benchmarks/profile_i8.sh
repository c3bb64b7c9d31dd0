#!/bin/bash
# Round 5: counters of the int8 candidate pass at config 2's shape (256 queries x top-100, 10M x 768) next to the half hi pass
# (OTT_HI_FMT=1), each --pmc set in its own run (never combined with other trace domains); kernel trace for the times.
# Run on the GPU box: bash benchmarks/profile_i8.sh <outdir>
R=$(cd "$(dirname "$0")/.." && pwd)
O=${1:-$R/gpurun_out/prof_i8}
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
cd "$R"
for MODE in i8 half; do
  unset OTT_HI_FMT
  [ $MODE = half ] && export OTT_HI_FMT=1
  for NQ in 256 1; do
    rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_${MODE}_$NQ" -- python3 benchmarks/mfma_batch.py $NQ > "$O/stats_${MODE}_$NQ.log" 2>&1
  done
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$O/mfma_$MODE" -- python3 benchmarks/mfma_batch.py 256 > "$O/mfma_$MODE.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch_$MODE" -- python3 benchmarks/mfma_batch.py 256 > "$O/fetch_$MODE.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch1_$MODE" -- python3 benchmarks/mfma_batch.py 1 > "$O/fetch1_$MODE.log" 2>&1
done
unset OTT_HI_FMT
python3 - "$O" <<'PY'
import csv, collections, glob, sys
O = sys.argv[1]
def load(pat):
    d = collections.defaultdict(lambda: collections.defaultdict(float)); meta = {}
    for f in glob.glob(pat):
        for r in csv.DictReader(open(f)):
            if "mfma_score" not in r["Kernel_Name"]: continue
            d[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            meta[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    return d, meta
for mode in ("i8", "half"):
    d, meta = load(O + "/mfma_%s/*/*counter_collection.csv" % mode)
    if meta:
        big = max(meta, key=lambda k: meta[k][0]); x = d[big]; cyc = x["GRBM_GUI_ACTIVE"] / 8
        print("%s, 256 queries: largest dispatch %.3f ms (%s), %.2f GHz, MFMA busy %.1f %% of SIMD cycles" % (mode, meta[big][0] / 1e6, meta[big][1][:60], cyc / meta[big][0], 100 * x["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4)))
    for tag, nq in (("fetch", 256), ("fetch1", 1)):
        d, meta = load(O + "/%s_%s/*/*counter_collection.csv" % (tag, mode))
        if meta:
            tot = sum(v["FETCH_SIZE"] for v in d.values()) / 3  # three batches per run
            print("    %d queries: FETCH_SIZE over one batch's scoring dispatches: %.0f KiB reported = %.2f GB (x 1024 x 2, gfx950)" % (nq, tot, tot * 2048 / 1e9))
for mode in ("i8", "half"):
    for nq in (256, 1):
        for f in glob.glob(O + "/stats_%s_%d/*/*kernel_stats.csv" % (mode, nq)):
            for r in csv.DictReader(open(f)):
                if any(k in r["Name"] for k in ("mfma_score", "select_kernel", "finalize", "i8_rows", "hi_rows")):
                    print("  %s nq=%d  %-70s calls %s  avg %.1f us  total %.3f ms" % (mode, nq, r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
find "$O" -name "*.csv" | head -40
