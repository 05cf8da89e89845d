#!/bin/bash
# per-kernel average times of one bench invocation: tools/kstats.sh <bench args>
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_kstats
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --tile-path-tiles 0 "$@" > $OUT/bench.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/trace/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if 'nyxhip' in r['Name']:
        print(r['Name'][:80].ljust(80), r['Calls'], 'avg_ms', round(float(r['AverageNs'])/1e6,3))
PY
