#!/bin/bash
# Diagnostic libraries whose grey-depth-64 feature pass (glcm_features_wave64_v2) ends after phase k = 0 (skipped) .. 5:
#   gpurun_scratch/libg16_<k>.so;  time them with  tools/g16_phases.sh  (differences of consecutive k = the phases).
cd $(dirname $0)/../nyxus_amd/csrc
mkdir -p ../../gpurun_scratch/objg16
for k in 0 1 2 3 4 5; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -w -DNYX_G16_EXIT=$k -c -o ../../gpurun_scratch/objg16/rf_$k.o roi_features.hip &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../gpurun_scratch/libtex_$k.so ../../gpurun_scratch/objg16/rf_$k.o $(ls obj/*.o | grep -v roi_features.o) ) &
done
wait
ls -la ../../gpurun_scratch/libtex_*.so
