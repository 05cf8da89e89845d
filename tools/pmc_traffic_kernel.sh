#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, KB per dispatch) of the kernels matching a substring: tools/pmc_traffic_kernel.sh <substr> <bench args...>
# (separate --pmc passes, no tracing; FETCH_SIZE is doubled for gfx950 as MI355X_MICROARCH.md prescribes)
export TMPDIR=/tmp
K=$1; shift
for C in FETCH_SIZE WRITE_SIZE; do
  OUT=$PWD/gpurun_out/pmc_traffic_$C; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc $C --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras --tile-path-tiles 0 "$@" > $OUT/log.txt 2>&1
done
python3 - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(list))
for C in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob("$PWD/gpurun_out/pmc_traffic_%s/**/*counter_collection.csv" % C, recursive=True):
        for r in csv.DictReader(open(f)):
            if "$K" in r.get('Kernel_Name',''): acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,m in acc.items():
    f=sum(m['FETCH_SIZE'])/max(len(m['FETCH_SIZE']),1); w=sum(m['WRITE_SIZE'])/max(len(m['WRITE_SIZE']),1)
    print(k, 'dispatches', len(m['FETCH_SIZE']), 'FETCH_GB(x2 gfx950)', round(2*f*1024/1e9,3), 'WRITE_GB', round(w*1024/1e9,3))
PY
