#!/bin/bash
# Vector / scalar / LDS instructions per wave of the metric kernel up to each phase boundary (tools/headline_phase_libs.sh builds the
# libraries): one rocprofv3 --pmc pass per library, counters alone.  Output: gpurun_out/headline_phases.txt
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/headline_phases; rm -rf $OUT; mkdir -p $OUT
for lib in $(ls $PWD/gpurun_scratch/libexit_*.so | sort -t_ -k2 -n) ""; do
  if [ -z "$lib" ]; then unset NYXHIP_LIB; tag=full; else export NYXHIP_LIB=$lib; tag=$(basename $lib .so); fi
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $OUT/$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras --tile-path-tiles 0 > $OUT/$tag.log 2>&1 < /dev/null
done
python3 - <<PY | tee $PWD/gpurun_out/headline_phases.txt
import csv, glob, os, re
from collections import defaultdict
rows = []
for d in sorted(glob.glob("$OUT/*/"), key=lambda p: (p.rstrip('/').split('/')[-1] == 'full', int(re.sub(r'\D', '', p.rstrip('/').split('/')[-1]) or 0))):
    tag = d.rstrip('/').split('/')[-1]
    acc = defaultdict(list)
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if 'roi_features_kernel_occ8' in r.get('Kernel_Name', ''):
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    if not acc: continue
    avg = {c: sum(v) / len(v) for c, v in acc.items()}
    w = avg.get('SQ_WAVES', 1) or 1
    print(tag.ljust(14), ' '.join(f"{c[3:]} {avg[c] / w:9.1f}" for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_WAVE_CYCLES') if c in avg))
PY
