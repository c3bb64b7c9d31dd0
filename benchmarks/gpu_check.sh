export OTT_LIB_PATH=$PWD/otters_amd/csrc/libotters_hip_dbg.so
for abl in 2 10 6 14; do echo "--- ablation $abl (2 = DMA only, 10 = DMA only + 16 more pieces per wave in flight; 6/14 same without fragment reads)"; OTT_MFMA_ABL=$abl OTT_MFMA_DEBUG=1 OTT_HI256=1 python benchmarks/mfma_batch.py 256 2>&1 | grep "hi256 dbg" | tail -1 | cut -c1-120; done
