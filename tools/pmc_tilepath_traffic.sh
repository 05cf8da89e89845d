#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_tilepath2; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 tools/two_ctx_probe.py 1000 > $OUT/log_$c.txt 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/two_ctx_probe.py 1000 > $OUT/log_trace.txt 2>&1
python3 - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r.get('Kernel_Name','')
        if 'nyxhip' in n:
            acc[n[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for n,m in acc.items():
    print(n, {k: '%.3g GB (n=%d)' % (sum(v)/len(v)*1024/1e9, len(v)) for k,v in sorted(m.items())})
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'nyxhip' in r['Name']: print(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e6)
PY
