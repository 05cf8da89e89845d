#!/bin/bash
# kernel trace + stats only (no PMC passes): tools/trace_only.sh <tag> [bench args]
export TMPDIR=/tmp
TAG=$1; shift
OUT=$PWD/gpurun_out/trace_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check "$@" > $OUT/bench.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print('%-90s calls %5s avg %10.1f us  min %10.1f  max %10.1f' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
tail -1 $OUT/bench.log | cut -c1-400
