export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_g64; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-extras --tile-path-tiles 0 --gray-depth 64 > $OUT/bench.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    seen=set()
    for r in csv.DictReader(open(f)):
        if 'nyxhip' in r['Kernel_Name'] and r['Kernel_Name'] not in seen:
            seen.add(r['Kernel_Name']); print(r['Kernel_Name'][:60], 'vgpr',r.get('VGPR_Count'),'lds',r.get('LDS_Block_Size'),'grid',r.get('Grid_Size_X'), 'ms', (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
PY
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras --tile-path-tiles 0 --gray-depth 64 > $OUT/pmc.log 2>&1
python3 - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'roi_features' in r.get('Kernel_Name',''): acc[r['Counter_Name']].append(float(r['Counter_Value']))
w=sum(acc['SQ_WAVES'])/len(acc['SQ_WAVES'])
print({k: round(sum(v)/len(v)/w,1) for k,v in acc.items()})
PY
