#!/bin/bash
# Early-exit builds of roi_features.hip for the per-phase instruction budget of the metric kernel (results are wrong by design):
#   gpurun_scratch/libexit_<k>.so ends roi_features_kernel at STAMP(k).  Measure with tools/headline_phase_insts.sh on the GPU box.
cd $(dirname $0)/../nyxus_amd/csrc
mkdir -p ../../gpurun_scratch/objexit
build() {
  k=$1
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -w -DNYX_EXIT_AT=$k -c -o ../../gpurun_scratch/objexit/rf_$k.o roi_features.hip &&
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../gpurun_scratch/libexit_$k.so ../../gpurun_scratch/objexit/rf_$k.o $(ls obj/*.o | grep -v roi_features.o)
}
for grp in "0 1 2 3" "4 5 6 7" "8 10 11 12"; do
  for k in $grp; do build $k & done
  wait
done
ls ../../gpurun_scratch/libexit_*.so
