#!/bin/bash
# cumulative counters of roi_moments_kernel for the early-exit builds gpurun_scratch/libmom_<k>.so (-DNYX_MOM_EXIT_AT=k)
export TMPDIR=/tmp
for lib in $(ls $PWD/gpurun_scratch/libmom_*.so | sort -t_ -k2 -n) $PWD/nyxus_amd/libnyxhip.so; do
  OUT=$PWD/gpurun_out/pmc_mom; rm -rf $OUT; mkdir -p $OUT
  export NYXHIP_LIB=$lib
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-check --no-extras --tile-path-tiles 0 --families 3072 > $OUT/log.txt 2>&1 < /dev/null
  python3 - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'roi_moments' in r.get('Kernel_Name',''): acc[r['Counter_Name']].append(float(r['Counter_Value']))
m={k: sum(v)/len(v) for k,v in acc.items()}
w=m.get('SQ_WAVES',1)
print("$(basename $lib)", {k: round(v/w,1) for k,v in sorted(m.items())})
PY
done
