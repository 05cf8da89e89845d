#!/bin/bash
# time the metric kernel with each ablation build found in gpurun_scratch/ (diagnostic; results of those builds are wrong by design)
for lib in "" $PWD/gpurun_scratch/libabl_*.so; do
  if [ -z "$lib" ]; then unset NYXHIP_LIB; else export NYXHIP_LIB=$lib; fi
  timeout 250 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check --tile-path-tiles 0 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename "$lib")', d['roofline']['kernel_ms'])"
done
