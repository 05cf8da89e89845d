#!/bin/bash
# time the metric kernel with each diagnostic build found in gpurun_scratch/ (results of those builds are wrong by design)
for lib in "" $PWD/gpurun_scratch/lib*.so; do
  if [ -z "$lib" ]; then unset NYXHIP_LIB; else export NYXHIP_LIB=$lib; fi
  timeout 250 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check --tile-path-tiles 0 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename "$lib")', round(d['roofline']['kernel_ms'],3))"
done
