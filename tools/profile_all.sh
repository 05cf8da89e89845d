#!/bin/bash
# Every reduce kernel on the metric ROIs (all twelve families), one rocprofv3 pass per counter group:
#   tools/profile_all.sh <tag> [bench args]      -> gpurun_out/prof_all_<tag>/summary.txt
# Kernel trace + stats in one run; each --pmc group in a run of its own (never combined with trace domains).
export TMPDIR=/tmp
TAG=${1:-r02}; shift
OUT=$PWD/gpurun_out/prof_all_$TAG; rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras --tile-path-tiles 0 --families 4095 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 bench.py $ARGS > $OUT/pmc$i.log 2>&1
done
python3 - <<PY > $OUT/summary.txt
import csv,glob,re
from collections import defaultdict
short=lambda n: re.sub(r'\(.*','',n).replace('void ','').replace('nyxhip::','')[:60]
print("== rocprofv3 --kernel-trace --stats, bench.py $ARGS (196 k ROIs of the metric workload per launch) ==")
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'nyxhip' in r['Name']:
            print("%-62s calls %3s  avg %9.3f ms  min %9.3f  max %9.3f" % (short(r['Name']), r['Calls'], float(r['AverageNs'])/1e6, float(r['MinNs'])/1e6, float(r['MaxNs'])/1e6))
print("== per-dispatch resources (kernel trace) ==")
seen=set()
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=short(r['Kernel_Name'])
        if 'nyxhip' in r['Kernel_Name'] and n not in seen:
            seen.add(n); print("%-62s vgpr %4s sgpr %4s lds %7s scratch %5s grid %9s wg %4s" % (n, r.get('VGPR_Count'), r.get('SGPR_Count'), r.get('LDS_Block_Size'), r.get('Scratch_Size'), r.get('Grid_Size_X'), r.get('Workgroup_Size_X')))
print("== PMC, mean per dispatch (separate --pmc passes) ==")
acc=defaultdict(lambda: defaultdict(list))
for f in glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'nyxhip' in r.get('Kernel_Name',''):
            acc[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k,m in sorted(acc.items()):
    w=sum(m['SQ_WAVES'])/len(m['SQ_WAVES']) if 'SQ_WAVES' in m else 0
    print(k, "waves/dispatch %.0f" % w)
    for c,v in sorted(m.items()):
        mean=sum(v)/len(v)
        extra = "  per wave %.1f" % (mean/w) if w and c.startswith('SQ_') and c!='SQ_WAVES' else ""
        if c=='FETCH_SIZE': extra="  = %.4g B (x2 gfx950 correction for wide coalesced reads: %.4g B)" % (mean*1024, 2*mean*1024)
        if c=='WRITE_SIZE': extra="  = %.4g B" % (mean*1024)
        print("    %-24s %14.5g%s" % (c, mean, extra))
PY
cat $OUT/summary.txt
