#!/bin/bash
# Hardware counters of the nyxhip kernels of any command, one rocprofv3 pass per counter group (counters are collected in runs of
# their own: no trace domains beside --pmc on this pool).
#   tools/pmc.sh <set> <kernel substring | all> [tag] -- <command ...>
#   sets:  mix      instruction mix + occupancy         lds      LDS pipe (conflicts, atomics)
#          sq       busy / wait breakdown (4 passes)    traffic  HBM bytes (FETCH_SIZE x 2 on gfx950, WRITE_SIZE; MI355X_MICROARCH.md)
# The program after `--` is started directly by the profiler (python3 <script> ..., never a shell or env wrapper).
# Examples:  tools/pmc.sh mix roi_features -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras --tile-path-tiles 0
#            tools/pmc.sh traffic all tilepath -- python3 tools/two_ctx_probe.py 1000
export TMPDIR=/tmp
SET=$1; K=$2; shift 2
TAG=$SET
if [ "$1" != "--" ]; then TAG=$1; shift; fi
shift
case $SET in
  mix) GROUPS_=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR") ;;
  lds) GROUPS_=("SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS") ;;
  sq) GROUPS_=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY"
               "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY"
               "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_IFETCH SQ_WAIT_INST_ANY"
               "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64") ;;
  traffic) GROUPS_=("FETCH_SIZE" "WRITE_SIZE") ;;
  *) echo "unknown counter set $SET" >&2; exit 2 ;;
esac
OUT=$PWD/gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- "$@" > $OUT/g$i.log 2>&1 < /dev/null
done
python3 - <<PY | tee $PWD/gpurun_out/pmc_$TAG.txt
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r.get('Kernel_Name', '')
        if ('nyxhip' in n or 'class_' in n or 'tile_' in n) and ("$K" == "all" or "$K" in n):
            acc[n[:90]][r['Counter_Name']].append(float(r['Counter_Value']))
for n, m in sorted(acc.items()):
    avg = {c: sum(v) / len(v) for c, v in m.items()}
    if "$SET" == "traffic":
        print(n, 'dispatches', len(m.get('FETCH_SIZE', [])), 'FETCH_GB(x2 gfx950)', round(2 * avg.get('FETCH_SIZE', 0) * 1024 / 1e9, 4), 'WRITE_GB', round(avg.get('WRITE_SIZE', 0) * 1024 / 1e9, 4))
    else:
        w = avg.get('SQ_WAVES', 1) or 1
        print(n, 'waves', w, {c: round(v / w, 1) for c, v in sorted(avg.items()) if c != 'SQ_WAVES'})
PY
