#!/bin/bash
# Evidence behind profiles/<round>/configs_*: the BASELINE configs un-profiled, their rocprofv3 kernel stats, and the
# MFMA-pipe counters of the C2 batch (each --pmc set in its own run, never combined with other trace domains).
# Run on the GPU box: bash benchmarks/profile_configs.sh <outdir>
R=$(cd "$(dirname "$0")/.." && pwd)
O=${1:-$R/gpurun_out/prof_cfg}
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
cd "$R"
python3 benchmarks/run_configs.py > "$O/configs_plain.log" 2>&1
python3 benchmarks/nq_sweep.py > "$O/nq_sweep.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 benchmarks/run_configs.py c1 head c2 c3 > "$O/configs_stats.log" 2>&1
# C2 counters for the three candidate passes: default cascade (hi pass), split pass alone, f32 matrix pipe
for MODE in hi split f32pipe; do
  unset OTT_NO_HI_PASS OTT_MFMA_F32
  [ $MODE = split ] && export OTT_NO_HI_PASS=1
  [ $MODE = f32pipe ] && export OTT_MFMA_F32=1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$O/mfma_$MODE" -- python3 benchmarks/mfma_batch.py 256 > "$O/c2_mfma_$MODE.log" 2>&1
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d "$O/waits_$MODE" -- python3 benchmarks/mfma_batch.py 256 > "$O/c2_waits_$MODE.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch_$MODE" -- python3 benchmarks/mfma_batch.py 256 > "$O/c2_fetch_$MODE.log" 2>&1
done
unset OTT_NO_HI_PASS OTT_MFMA_F32
python3 - "$O" <<'PY'
import csv, collections, glob, sys
O = sys.argv[1]
def load(pat):
    d = collections.defaultdict(lambda: collections.defaultdict(float)); meta = {}
    for f in glob.glob(pat):
        for r in csv.DictReader(open(f)):
            if "mfma_score" not in r["Kernel_Name"]: continue
            d[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            meta[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return d, meta
for mode in ("hi", "split", "f32pipe"):
    d, meta = load(O + "/mfma_%s/*/*counter_collection.csv" % mode)
    if not meta: continue
    big = max(meta, key=lambda k: meta[k]); x = d[big]; cyc = x["GRBM_GUI_ACTIVE"] / 8
    print("%s: largest dispatch %.3f ms, %.2f GHz, MFMA busy %.1f %% of SIMD cycles" % (mode, meta[big] / 1e6, cyc / meta[big], 100 * x["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4)))
    d, meta = load(O + "/waits_%s/*/*counter_collection.csv" % mode)
    big = max(meta, key=lambda k: meta[k]); y = d[big]
    print("   ", {k: round(v / y["SQ_WAVE_CYCLES"], 4) for k, v in y.items() if k != "SQ_WAVE_CYCLES"})
    d, meta = load(O + "/fetch_%s/*/*counter_collection.csv" % mode)
    tot = sum(v["FETCH_SIZE"] for k, v in d.items()) / 3  # three batches per run
    print("    FETCH_SIZE over one batch's scoring dispatches: %.0f KB reported" % tot)
PY
grep "^|" "$O/configs_plain.log"
tail -18 "$O/nq_sweep.log"
find "$O" -name "*.csv" | head
