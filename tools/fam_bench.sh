#!/bin/bash
for f in "$@"; do
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check --no-extras --tile-path-tiles 0 --families $f 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('families', $f, 'ms', round(d['ms_per_step'],3), 'ROIs/s', round(d['value']))"
done
