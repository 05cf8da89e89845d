#!/bin/bash
# instruction-mix counters for roi_features_kernel (own pass, no trace domains)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_mix_$1
mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/a -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --tile-path-tiles 0 > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_F64 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --tile-path-tiles 0 > $OUT/b.log 2>&1
python3 - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'roi_features' in r.get('Kernel_Name',''): acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(f"{k}: {sum(v)/len(v):.5g}  per wave {sum(v)/len(v)/784000:.1f}")
PY
tail -3 $OUT/b.log | cut -c1-200
