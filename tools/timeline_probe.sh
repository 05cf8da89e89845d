#!/bin/bash
# kernel timeline of the LAST call of a command under rocprofv3's kernel trace (start / end in ms relative to the call's first kernel, queue id):
#   tools/timeline_probe.sh python3 tools/size_legs.py --families 4095 --no-sweep
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/timeline_probe
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- "$@" > $OUT/cmd.log 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')))
rows.sort()
# the last call: walk back from the end to the last class_count_kernel
idx = max(i for i, r in enumerate(rows) if 'class_count_kernel' in r[2])
t0 = rows[idx][0]
for s, e, n, q in rows[idx:]:
    if (e - s) > 150000:
        n = n.replace('nyxhip::', '').replace('(anonymous namespace)::', '').replace('void ', '')
        print(f"{(s - t0) / 1e6:8.3f} -> {(e - t0) / 1e6:8.3f} ms  q{q:>3}  {n[:80]}")
print('call span ms', (max(r[1] for r in rows[idx:]) - t0) / 1e6)
PY
