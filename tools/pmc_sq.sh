#!/bin/bash
# SQ busy / wait breakdown of the metric kernel (one pass per counter group): tools/pmc_sq.sh <tag> [bench args]
export TMPDIR=/tmp
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_sq_$TAG; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_IFETCH SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --tile-path-tiles 0 "$@" > $OUT/g$i.log 2>&1
done
python3 - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r.get('Kernel_Name','')
        k='roi_features' if 'roi_features' in n else 'glcm_features' if 'glcm_features' in n else None
        if k and int(r.get('Grid_Size') or 0) > 1000000: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,m in acc.items():
    w=sum(m['SQ_WAVES'])/len(m['SQ_WAVES']) if 'SQ_WAVES' in m else 1
    print(k,'waves',w)
    for c,v in sorted(m.items()): print('   %-28s %14.4g  per wave %10.1f' % (c, sum(v)/len(v), sum(v)/len(v)/w))
PY
