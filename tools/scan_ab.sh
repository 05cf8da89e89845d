#!/bin/bash
# A/B of tile-scan variants in gpurun_scratch (libscan_*.so) on the 1000-tile path: kernel time of tile_scan_kernel + the leg's rate
export TMPDIR=/tmp
for lib in "" $PWD/gpurun_scratch/libscan_*.so; do
  OUT=$PWD/gpurun_out/scan_ab; rm -rf $OUT; mkdir -p $OUT
  NYXHIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras --tile-path-tiles 1000 > $OUT/log.txt 2>&1
  python3 - <<PY
import csv,glob,json
name="$(basename "${lib:-default}")"
mx=0
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'tile_scan_kernel<unsigned int, unsigned int>' in r['Kernel_Name']:
            mx=max(mx,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
try:
    d=json.loads(open("$OUT/log.txt").read().strip().splitlines()[-1]); tp=d['tile_path']['value']/1e6
except Exception as e: tp=-1
print(name, 'scan max ms', round(mx,3), 'tile path M ROIs/s', round(tp,2))
PY
done
