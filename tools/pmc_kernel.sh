#!/bin/bash
# SQ instruction mix / occupancy of one kernel (substring match) for a bench invocation: tools/pmc_kernel.sh <substr> <bench args...>
export TMPDIR=/tmp
K=$1; shift
OUT=$PWD/gpurun_out/pmc_kernel; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --tile-path-tiles 0 "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$K" in r.get('Kernel_Name',''): acc[r['Counter_Name']].append(float(r['Counter_Value']))
m={k: sum(v)/len(v) for k,v in acc.items()}
w=m.get('SQ_WAVES',1)
print("$K", {k: round(v/w,1) for k,v in sorted(m.items())}, "waves", w)
PY
