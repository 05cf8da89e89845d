#!/bin/bash
# Diagnostic libraries whose 8-level GLCM feature routine (glcm_features_wave8) ends after phase k: 1 marginals, 2 cell pass, 3 marginal terms
# + reductions (the rest: the closing formulas of lane 0).  gpurun_scratch/libg8_<k>.so; counters:  tools/g8_phase_insts.sh
cd $(dirname $0)/../nyxus_amd/csrc
mkdir -p ../../gpurun_scratch/objg8
for k in 1 2 3; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -w -DNYX_G8_EXIT=$k -c -o ../../gpurun_scratch/objg8/rf_$k.o roi_features.hip &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../gpurun_scratch/libg8_$k.so ../../gpurun_scratch/objg8/rf_$k.o $(ls obj/*.o | grep -v roi_features.o) ) &
done
wait
ls -la ../../gpurun_scratch/libg8_*.so
