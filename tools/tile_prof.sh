#!/bin/bash
# Kernel trace of the fused tile path (run on the GPU box): tools/tile_prof.sh <tag> <tiles>
export TMPDIR=/tmp
TAG=${1:-tile}; N=${2:-1000}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 1 --warmup 1 --tiles 16 --no-cpu-baseline --no-check --no-extras --tile-path-tiles $N > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['tile_path'])"
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/trace/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if 'nyxhip' in r['Name']:
        print(r['Name'][:60].ljust(60), r['Calls'], 'avg_us', float(r['AverageNs'])/1e3, 'max_us', float(r['MaxNs'])/1e3)
PY
