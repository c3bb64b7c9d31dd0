mkdir -p gpurun_out
export OTT_LIB_PATH=$GRAFT_REPO_ROOT/otters_amd/csrc/libotters_hip_dbg.so OTT_HI256=1 OTT_MFMA_DEBUG=1
for a in 0 1 4 5 6 7; do echo "== ABL $a"; OTT_MFMA_ABL=$a python benchmarks/mfma_batch.py 256 2>&1 | grep -E "hi256 dbg" | tail -2; done
