#!/bin/bash
# A/B: current library vs gpurun_scratch/libhead.so on a few metric-kernel configurations
for args in "--gray-depth 8" "--gray-depth 64" "--gray-depth 8 --families 1" "--gray-depth 64 --families 1" "--gray-depth 8 --families 2"; do
  for lib in "" $PWD/gpurun_scratch/libhead.so; do
    if [ -z "$lib" ]; then unset NYXHIP_LIB; tag=new; else export NYXHIP_LIB=$lib; tag=head; fi
    timeout 250 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check --tile-path-tiles 0 $args 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$args', '$tag', round(d['roofline']['kernel_ms'],3))"
  done
done
