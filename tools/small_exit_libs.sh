#!/bin/bash
# Early-exit builds of roi_small.hip (per-phase instruction budget of the wave-per-ROI kernel; results are wrong by design):
#   gpurun_scratch/libsmall_<k>.so, k = 1 load | 2 sums, central sums, their outputs | 3 sort | 4 bin bounds | 5 percentiles, entropy | 6 median, mode | 7 robust statistics
cd $(dirname $0)/../nyxus_amd/csrc
mkdir -p ../../gpurun_scratch/objsmall
for k in 1 2 3 4 5 6 7; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -w -DNYX_SMALL_EXIT=$k -c -o ../../gpurun_scratch/objsmall/rs_$k.o roi_small.hip &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../gpurun_scratch/libsmall_$k.so ../../gpurun_scratch/objsmall/rs_$k.o $(ls obj/*.o | grep -v roi_small.o) ) &
done
wait
ls ../../gpurun_scratch/libsmall_*.so
