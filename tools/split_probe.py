"""Experiment: the metric workload as two concurrent launches (INTENSITY on one stream, GLCM on another, writing
disjoint column ranges of the same table) versus the fused kernel.  Diagnostic only."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from nyxus_amd import _abi, _lib

dev = torch.device("cuda", 0)
tiles = 1000
r = 30
yy, xx = np.mgrid[-r:r + 1, -r:r + 1]
m = (xx * xx + yy * yy) <= r * r
y, x = np.nonzero(m)
o = np.lexsort((y, x))
px, py = x[o].astype(np.uint16), y[o].astype(np.uint16)
npx = len(px); n_roi = tiles * 196
inten = torch.randint(1, 4096, (n_roi * npx,), device=dev, dtype=torch.int32)
X = torch.from_numpy(px.view(np.int16)).to(dev).repeat(n_roi)
Y = torch.from_numpy(py.view(np.int16)).to(dev).repeat(n_roi)
off = torch.arange(0, n_roi + 1, device=dev, dtype=torch.int64) * npx
bw = torch.full((n_roi,), 61, device=dev, dtype=torch.int32); bh = bw.clone()
iv = inten.view(n_roi, npx); mn = iv.min(1).values.contiguous(); mx = iv.max(1).values.contiguous()
lab = torch.arange(n_roi, device=dev, dtype=torch.int32)
cb = _abi.Batch(); cb.n_roi = n_roi; cb.roi_label = lab.data_ptr(); cb.px_offset = off.data_ptr(); cb.x = X.data_ptr(); cb.y = Y.data_ptr()
cb.inten = inten.data_ptr(); cb.bbox_w = bw.data_ptr(); cb.bbox_h = bh.data_ptr(); cb.min_inten = mn.data_ptr(); cb.max_inten = mx.data_ptr()
cb.slide_min = None; cb.slide_max = None; cb.memory = _abi.MEM_DEVICE; cb.max_px = npx; cb.max_bbox_area = 61 * 61
cb.max_inten_range = int((mx - mn).max().item()); cb.max_bbox_side = 61
s = _abi.default_settings(8)
out = torch.empty((n_roi, 185), dtype=torch.float64, device=dev)
out2 = torch.empty((n_roi, 185), dtype=torch.float64, device=dev)
c0 = _lib.Context(0); c1 = _lib.Context(0); c2 = _lib.Context(0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
c1.set_stream(s1.cuda_stream); c2.set_stream(s2.cuda_stream)

def fused():
    c0.featurize_device_async(cb, 3, s, out.data_ptr(), 185)

def split():
    c1.featurize_device_async(cb, 1, s, out2.data_ptr(), 185)
    c2.featurize_device_async(cb, 2, s, out2.data_ptr() + 36 * 8, 185)

for name, fn in (("fused", fused), ("split", split), ("fused", fused), ("split", split)):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    print(name, (time.perf_counter() - t) / 10 * 1e3, "ms")
print("same:", torch.equal(out, out2) or float((out - out2).abs().max()))
