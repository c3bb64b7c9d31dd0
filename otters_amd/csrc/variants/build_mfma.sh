#!/bin/bash
# Experiment builds of the PRODUCT library in which only ott_mfma.hip differs (knobs OTT_X_*); the other objects come from _obj.
# usage: variants/build_exact.sh name "-DOTT_X_...=.." [name "flags" ...]   (run from otters_amd/csrc; builds in parallel)
cd "$(dirname "$0")/.."
pids=()
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  (
    /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function $flags -c ott_mfma.hip -o variants/mfma_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_$name.so $(ls _obj/*.o | grep -v ott_mfma.o) variants/mfma_$name.o -ldl && echo "built $name"
  ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
