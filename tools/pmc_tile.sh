#!/bin/bash
# instruction / LDS counters of the tile path's kernels: tools/pmc_tile.sh [tiles]
export TMPDIR=/tmp
N=${1:-1000}
OUT=$PWD/gpurun_out/pmc_tile; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-check --no-extras --tile-path-tiles $N > $OUT/log.txt 2>&1 < /dev/null
python3 - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'nyxhip' in r.get('Kernel_Name',''): acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,mm in acc.items():
    m={c: sum(v)/len(v) for c,v in mm.items()}
    w=m.get('SQ_WAVES',1)
    print(k, {c: round(v/w,1) for c,v in sorted(m.items())}, 'waves', w)
PY
