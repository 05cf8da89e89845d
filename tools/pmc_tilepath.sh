#!/bin/bash
# SQ instruction counts of the reduce kernels inside the device tile path (window mode): tools/pmc_tilepath.sh
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_tilepath; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d $OUT -- python3 tools/two_ctx_probe.py 1000 > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r.get('Kernel_Name','')
        if 'nyxhip' in n and int(r.get('Grid_Size') or 0) >= 25000000:
            acc[n[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for n,m in acc.items():
    w=sum(m['SQ_WAVES'])/len(m['SQ_WAVES'])
    print(n, 'waves', w, {k: round(sum(v)/len(v)/w,1) for k,v in sorted(m.items()) if k!='SQ_WAVES'})
PY
