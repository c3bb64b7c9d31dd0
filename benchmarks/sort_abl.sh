# timing ablations of the radix-sort pass kernel (results are WRONG under any of them): which phase a pass spends its time in
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for A in 0 1 2 3 4 8 15; do
  OTT_MFMA_ABL=$A rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sortabl_$A -- python3 benchmarks/sort_probe.py > gpurun_out/sortabl_$A.log 2>&1
  f=$(ls -t $(find gpurun_out/sortabl_$A -name "*kernel_stats.csv") | head -1)
  echo "abl=$A $(grep rs_pass $f | cut -d, -f3,4,5)"
done
