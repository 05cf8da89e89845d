#!/bin/bash
# SQ_INSTS_VALU / SALU / LDS of the metric kernel for each early-exit build in gpurun_scratch (diagnostic)
export TMPDIR=/tmp
for lib in $PWD/gpurun_scratch/libexit_*.so; do
  OUT=$PWD/gpurun_out/pmc_exit/$(basename $lib .so)
  mkdir -p $OUT
  NYXHIP_LIB=$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras --tile-path-tiles 0 > $OUT/log.txt 2>&1
  python3 - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'roi_features' in r.get('Kernel_Name',''): acc[r['Counter_Name']].append(float(r['Counter_Value']))
print("$(basename $lib)", {k: round(sum(v)/len(v)/784000,1) for k,v in sorted(acc.items())})
PY
done
