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
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$O/mfma" -- python3 benchmarks/mfma_batch.py 256 > "$O/c2_mfma.log" 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d "$O/waits" -- python3 benchmarks/mfma_batch.py 256 > "$O/c2_waits.log" 2>&1
grep "^|" "$O/configs_plain.log"
tail -18 "$O/nq_sweep.log"
find "$O" -name "*.csv" | head
