export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_mc; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- python3 tools/tile_dtype_probe.py 64 > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob
ev=[]
for f in glob.glob("$OUT/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY '+r.get('Direction','')+' '+str(r.get('Size','') or r.get('Bytes',''))))
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'nyxhip' in r['Kernel_Name']: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40]))
ev.sort()
# show the last call of the first dtype config: find big copies
big=[e for e in ev if e[2].startswith('COPY') ]
print(len(ev), 'events;', len(big), 'copies')
t0=ev[0][0]
# print a window of 120 events from the middle of the uint32 runs
import itertools
start=None
cnt=0
for e in ev:
    if e[2].startswith('COPY') and ('HOST_TO_DEVICE' in e[2] or 'H2D' in e[2].upper()):
        cnt+=1
        if cnt==9: start=e[0]; break
if start:
    for e in ev:
        if start <= e[0] <= start+30_000_000:
            print("%9.3f ms  +%8.3f ms  %s" % ((e[0]-start)/1e6, (e[1]-e[0])/1e6, e[2]))
PY
