#!/bin/bash
# per-kernel times of any command under rocprofv3:  tools/ktrace.sh <tag> python3 <script> [args...]   (summary -> gpurun_out/ktrace_<tag>.txt)
export TMPDIR=/tmp
TAG=$1; shift
OUT=$PWD/gpurun_out/ktrace_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- "$@" > $OUT/cmd.log 2>&1
python3 - <<PY > $PWD/gpurun_out/ktrace_$TAG.txt
import csv,glob
for f in glob.glob("$OUT/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if 'nyxhip' in r['Name'] or 'class_' in r['Name']:
            print(r['Name'][:100].ljust(100), r['Calls'].rjust(5), 'avg_us', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))
PY
cat $PWD/gpurun_out/ktrace_$TAG.txt
