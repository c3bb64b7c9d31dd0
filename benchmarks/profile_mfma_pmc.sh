#!/bin/bash
# MFMA-pipe and wait counters of one batch size (default 64 queries): bash benchmarks/profile_mfma_pmc.sh <nq> <outdir>
R=$(cd "$(dirname "$0")/.." && pwd)
NQ=${1:-64}
O=${2:-$R/gpurun_out/prof_pmc_$NQ}
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
cd "$R"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$O/mfma" -- python3 benchmarks/mfma_batch.py $NQ > "$O/mfma.log" 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d "$O/waits" -- python3 benchmarks/mfma_batch.py $NQ > "$O/waits.log" 2>&1
python3 - "$O" <<'PY'
import csv, collections, glob, sys
O = sys.argv[1]
def load(pat):
    d = collections.defaultdict(lambda: collections.defaultdict(float)); meta = {}
    for f in glob.glob(pat):
        for r in csv.DictReader(open(f)):
            if "mfma_score" not in r["Kernel_Name"] and "hi256" not in r["Kernel_Name"]: continue
            d[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            meta[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return d, meta
d, meta = load(O + "/mfma/*/*counter_collection.csv")
big = max(meta, key=lambda k: meta[k]); x = d[big]; cyc = x["GRBM_GUI_ACTIVE"] / 8
print("largest dispatch: %.3f ms, %.2f GHz, MFMA busy %.1f %% of SIMD cycles" % (meta[big] / 1e6, cyc / meta[big], 100 * x["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4)))
d, meta = load(O + "/waits/*/*counter_collection.csv")
big = max(meta, key=lambda k: meta[k]); y = d[big]
print({k: round(v / y["SQ_WAVE_CYCLES"], 4) for k, v in y.items() if k != "SQ_WAVE_CYCLES"})
PY
