"""Family times on a resident batch of identical disks of radius R (box 2R+1): where the kernels' width-dependent paths switch.
usage: python3 tools/radius_probe.py R n_roi mask [mask ...]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from nyxus_amd import _abi, _lib

R = int(sys.argv[1]); n_roi = int(sys.argv[2]); masks = [int(m) for m in sys.argv[3:]]
dev = torch.device("cuda:0")
yy, xx = np.mgrid[-R:R + 1, -R:R + 1]
m = (xx * xx + yy * yy) <= R * R
y, x = np.nonzero(m); o = np.lexsort((y, x))
px, py, side = x[o].astype(np.uint16), y[o].astype(np.uint16), 2 * R + 1
npx = len(px); n_px = n_roi * npx
g = torch.Generator(device=dev); g.manual_seed(1)
inten = torch.randint(1, 4096, (n_px,), generator=g, device=dev, dtype=torch.int32)
X = torch.from_numpy(px.view(np.int16)).to(dev).repeat(n_roi); Y = torch.from_numpy(py.view(np.int16)).to(dev).repeat(n_roi)
off = torch.arange(0, n_roi + 1, device=dev, dtype=torch.int64) * npx
bw = torch.full((n_roi,), side, device=dev, dtype=torch.int32); bh = bw.clone()
iv = inten.view(n_roi, npx); mn = iv.min(dim=1).values.contiguous(); mx = iv.max(dim=1).values.contiguous()
labels = torch.arange(1, n_roi + 1, device=dev, dtype=torch.int32)
cb = _abi.Batch(); cb.n_roi = n_roi
cb.roi_label = labels.data_ptr(); cb.px_offset = off.data_ptr(); cb.x = X.data_ptr(); cb.y = Y.data_ptr(); cb.inten = inten.data_ptr()
cb.bbox_w = bw.data_ptr(); cb.bbox_h = bh.data_ptr(); cb.min_inten = mn.data_ptr(); cb.max_inten = mx.data_ptr()
cb.slide_min = None; cb.slide_max = None; cb.memory = _abi.MEM_DEVICE
cb.max_px = npx; cb.max_bbox_area = side * side; cb.max_inten_range = int((mx - mn).max().item()); cb.max_bbox_side = side
ctx = _lib.Context(0); s = _abi.default_settings(8)
for mask in masks:
    nc = ctx.n_columns(mask, s)
    out = torch.empty((n_roi, nc), dtype=torch.float64, device=dev)
    ctx.featurize_device_async(cb, mask, s, out.data_ptr(), nc); torch.cuda.synchronize()
    c0 = time.perf_counter()
    for _ in range(3): ctx.featurize_device_async(cb, mask, s, out.data_ptr(), nc)
    torch.cuda.synchronize(); ctx.sync()
    dt = (time.perf_counter() - c0) / 3
    print(f"R {R} box {side} px {npx} rois {n_roi} mask {mask}: {1e3 * dt:.2f} ms = {1e9 * dt / n_roi:.0f} ns/ROI", flush=True)
