"""BASELINE.json configs[4]: GABOR (8-orientation bank) + ZERNIKE2D on DSB2018-shaped ROIs, 1 MI355X.
ROIs = the four DSB2018 fixture shapes (tests/golden/reference_tests.json) replicated with seeded noise.
Prints one JSON line (informational; bench.py stays on the headline metric)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nyxus_amd import _abi, _lib
from oracle import pyoracle as po
from tests import fixtures, parity

n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
ref = fixtures.reference_tests()
rng = np.random.default_rng(5)
rois = []
for k in range(n_rep):
    for d in ref["dsb2018"]:
        r = fixtures.dsb_roi(d)
        v = r["inten"].astype(np.int64)
        v = np.where(v > 0, np.clip(v + rng.integers(-8, 9, len(v)), 1, 255), 0).astype(np.uint32)
        m = v > 0                      # ROI = the non-zero pixels of the fixture frame
        rois.append(dict(x=r["x"][m], y=r["y"][m], inten=v[m]))
b = _abi.batch_from_rois(rois)
s = _abi.default_settings(64)
s.gabor_n_filters = 8
for i in range(8):
    s.gabor_f0[i] = [4.0, 16.0, 32.0, 64.0][i % 4]
    s.gabor_theta[i] = np.pi * i / 8
mask = _abi.FAM_GABOR | _abi.FAM_ZERNIKE
ctx = _lib.Context(0)
ncol = ctx.n_columns(mask, s)
dev = {k: torch.from_numpy(getattr(b, k).view({2: np.int16, 4: np.int32, 8: np.int64}[getattr(b, k).dtype.itemsize])).cuda()
       for k in ("px_offset", "x", "y", "inten", "bbox_w", "bbox_h", "min_inten", "max_inten")}
out = torch.empty((b.n_roi, ncol), dtype=torch.float64, device="cuda")
cb = b.c_struct()
for k, t in dev.items():
    setattr(cb, k, t.data_ptr())
cb.slide_min = None; cb.slide_max = None; cb.memory = _abi.MEM_DEVICE
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(2):
    ctx.featurize_device_async(cb, mask, s, out.data_ptr(), ncol)
torch.cuda.synchronize()
t0 = time.perf_counter(); K = 5
for _ in range(K):
    ctx.featurize_device_async(cb, mask, s, out.data_ptr(), ncol)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
G = out[:400].cpu().numpy()
sub = _abi.batch_from_rois(rois[:400])
bad = parity.compare_tables(G, po.oracle_featurize(sub, mask, s), _lib.column_names(mask, s))
rec = {"workload": "GABOR (8 filters, 16x16) + ZERNIKE2D on DSB2018-shaped ROIs", "n_roi": b.n_roi, "mean_px": b.n_px / b.n_roi,
       "ROIs_per_s": b.n_roi / dt, "ms": 1e3 * dt, "parity_first_400": "ok" if not bad else bad[:3]}
if po.have_ref():
    tm = []; thr = os.cpu_count() or 1
    sub2 = _abi.batch_from_rois(rois[: 40 * thr])
    po.ref_featurize(sub2, mask, s, n_threads=thr, timing=tm)
    rec["cpu_reference_ROIs_per_s"] = sub2.n_roi / tm[0]; rec["cpu_threads"] = thr
print(json.dumps(rec))
