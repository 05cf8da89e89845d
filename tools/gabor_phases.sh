#!/bin/bash
# Vector / scalar / LDS / matrix instructions per wave of the Gabor kernel, phase by phase (cumulative), and the lengths of its
# recomputation lists: a diagnostic build (phase exits compiled in) under one rocprofv3 --pmc pass per phase.
#   build first (here, not on the GPU box):   make -C nyxus_amd/csrc EXTRA=-DNYXHIP_GABOR_PHASE_EXITS && cp nyxus_amd/libnyxhip.so gpurun_scratch/lib_diag.so
#                                             touch nyxus_amd/csrc/roi_shape.hip && make -C nyxus_amd/csrc          (the product build again)
#   then through gpurun:                      tools/gabor_phases.sh          (profiles/r05z_gabor_phases.txt was made this way)
export TMPDIR=/tmp
export NYXHIP_LIB=$PWD/gpurun_scratch/lib_diag.so
for p in 1 2 3 0; do
  export NYXHIP_DBG_PHASE=$p
  OUT=$PWD/gpurun_out/pmc_ph$p; rm -rf $OUT; mkdir -p $OUT
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_IFETCH --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras --tile-path-tiles 0 --families 32 > $OUT.log 2>&1 < /dev/null
  python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gabor_tiled_kernel<8' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
w = sum(acc['SQ_WAVES'])/len(acc['SQ_WAVES'])
print("after phase $p (0 = whole kernel)", {k: round(sum(v)/len(v)/w,1) for k,v in sorted(acc.items()) if k != 'SQ_WAVES'})
PY
done
NYXHIP_DBG_PHASE=6 python3 - <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
from nyxus_amd import _abi, _lib
ctx = _lib.Context(0)
s = _abi.default_settings(8, False)
rng = np.random.default_rng(0)
yy, xx = np.mgrid[-30:31, -30:31]; m = xx * xx + yy * yy <= 900
y0, x0 = np.nonzero(m)
rois = [dict(x=x0, y=y0, inten=rng.integers(1, 4096, len(x0)).astype(np.uint32)) for _ in range(2000)]
G = ctx.featurize_host(_abi.batch_from_rois(rois), _abi.FAM_GABOR, s)      # (phase 6: the output row carries the list lengths)
print("mean list lengths per ROI: low-pass candidates %.1f, band pixels of filters 1 .. 3: %.1f %.1f %.1f; max %s" % (*G.mean(axis=0), G.max(axis=0)))
PY
