"""Times nyxhip_featurize_tiles_v2 on host tiles of several element types (diagnostic; run under rocprofv3 --kernel-trace --stats)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nyxus_amd import _abi, _lib
from tests import synth
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = _lib.Context(0)
s = _abi.default_settings(8)
mask = 3
lab = synth.disk_label_tile()
rng = np.random.default_rng(1)
I = rng.integers(1, 4096, (nt, 1024, 1024), dtype=np.uint32)
L = np.broadcast_to(lab, (nt, 1024, 1024)).copy()
for ti, tl in ((np.uint32, np.uint32), (np.uint16, np.uint8), (np.uint16, np.uint16), (np.uint8, np.uint8)):
    a = (I % 250 + 1).astype(ti) if ti == np.uint8 else I.astype(ti)
    b = L.astype(tl)
    ctx.featurize_tiles_host(a, b, mask, s)
    t0 = time.perf_counter()
    for _ in range(3):
        r = ctx.featurize_tiles_host(a, b, mask, s)
    dt = (time.perf_counter() - t0) / 3
    print(np.dtype(ti).name, np.dtype(tl).name, "ms", round(1e3 * dt, 2), "rois", len(r[1]), "host GB/s", round((a.nbytes + b.nbytes) / dt / 1e9, 1), flush=True)
