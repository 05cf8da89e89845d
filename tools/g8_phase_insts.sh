#!/bin/bash
# cumulative instruction counts of glcm_features_kernel8's phases (tools/g8_exit_libs.sh builds), then the shipped library
export TMPDIR=/tmp
for lib in $PWD/gpurun_scratch/libg8_1.so $PWD/gpurun_scratch/libg8_2.so $PWD/gpurun_scratch/libg8_3.so $PWD/nyxus_amd/libnyxhip.so; do
  export NYXHIP_LIB=$lib
  echo "$(basename $lib): $(bash tools/pmc.sh mix glcm_features_kernel8 g8ph -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras --tile-path-tiles 0 2>&1 | tail -1 | cut -c1-330)"
done
