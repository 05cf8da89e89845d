#!/bin/bash
# idle time between consecutive kernels of the headline step under rocprofv3's kernel trace:  tools/gap_probe.sh [extra bench args]
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/gap_probe
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-check --no-extras --tile-path-tiles 0 "$@" > $OUT/cmd.log 2>&1
python3 - <<PY
import csv, glob, collections
rows = []
for f in glob.glob("$OUT/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]))
rows.sort()
gaps = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gaps[(n0, n1)].append((s1 - e0) / 1e3)
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    if len(v) >= 5:
        v2 = sorted(v)
        print(k[0].ljust(60), '->', k[1].ljust(60), 'n', len(v), 'median gap us', round(v2[len(v2) // 2], 1), 'min', round(v2[0], 1))
PY
