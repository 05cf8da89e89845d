#!/usr/bin/env python3
"""Size-dependent legs of the bench: `size_sweep` (homogeneous batches of disks at six sizes) and `mixed_sizes`
(log-normal radii 4..150 plus 1 % of 300..400-px boxes in ONE call).  The reference has no coupling between the
ROIs of a batch -- every worker thread takes ROIs of any size (/root/reference/src/nyx/parallel.h:23-42,
roi_cache.h:31-84) -- so the cost of a small ROI must not depend on what else shares its launch.

Used by bench.py (extras) and runnable on its own:  python tools/size_legs.py [--families 3] [--gray-depth 8]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_shape_cache = {}


def ellipse_cloud(a: int, b: int):
    """Column-major cloud (phase2_2d.cpp:655-656 scan order) of the ellipse with semi-axes a (x) and b (y)."""
    key = (a, b)
    if key not in _shape_cache:
        yy, xx = np.mgrid[-b:b + 1, -a:a + 1]
        m = (xx * xx) * (b * b) + (yy * yy) * (a * a) <= (a * a) * (b * b)
        y, x = np.nonzero(m)
        o = np.lexsort((y, x))
        _shape_cache[key] = (x[o].astype(np.uint16), y[o].astype(np.uint16))
    return _shape_cache[key]


class DeviceBatch:
    """A synthetic ROI batch resident in HBM: shapes from (a, b) ellipse templates, intensities U[lo, hi) seeded on the device."""

    def __init__(self, shapes, dev, seed=77, lo=1, hi=4096):
        import torch
        from nyxus_amd import _abi
        xs, ys, n = [], [], []
        for a, b in shapes:
            x, y = ellipse_cloud(a, b)
            xs.append(x); ys.append(y); n.append(len(x))
        self.shapes = list(shapes)
        self.n_px_roi = np.asarray(n, np.int64)
        off = np.concatenate([[0], np.cumsum(self.n_px_roi)])
        self.n_roi, self.n_px = len(n), int(off[-1])
        self.h_x, self.h_y, self.h_off = np.concatenate(xs), np.concatenate(ys), off.astype(np.uint64)
        self.h_bw = np.asarray([2 * a + 1 for a, _ in shapes], np.uint32)
        self.h_bh = np.asarray([2 * b + 1 for _, b in shapes], np.uint32)
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        self.inten = torch.randint(lo, hi, (self.n_px,), generator=g, device=dev, dtype=torch.int32)
        self.x = torch.from_numpy(self.h_x.view(np.int16)).to(dev)
        self.y = torch.from_numpy(self.h_y.view(np.int16)).to(dev)
        self.off = torch.from_numpy(off.astype(np.int64)).to(dev)
        self.bw = torch.from_numpy(self.h_bw.view(np.int32)).to(dev)
        self.bh = torch.from_numpy(self.h_bh.view(np.int32)).to(dev)
        lens = torch.from_numpy(self.n_px_roi).to(dev)
        f = self.inten.to(torch.float32)                                    # (values < 2^24: exact)
        self.mn = torch.segment_reduce(f, "min", lengths=lens).to(torch.int32).contiguous()
        self.mx = torch.segment_reduce(f, "max", lengths=lens).to(torch.int32).contiguous()
        self.labels = torch.arange(1, self.n_roi + 1, device=dev, dtype=torch.int32)
        cb = _abi.Batch()
        cb.n_roi = self.n_roi
        cb.roi_label = self.labels.data_ptr(); cb.px_offset = self.off.data_ptr()
        cb.x = self.x.data_ptr(); cb.y = self.y.data_ptr(); cb.inten = self.inten.data_ptr()
        cb.bbox_w = self.bw.data_ptr(); cb.bbox_h = self.bh.data_ptr()
        cb.min_inten = self.mn.data_ptr(); cb.max_inten = self.mx.data_ptr()
        cb.slide_min = None; cb.slide_max = None
        cb.memory = _abi.MEM_DEVICE
        cb.max_px = int(self.n_px_roi.max()); cb.max_bbox_area = int((self.h_bw.astype(np.int64) * self.h_bh).max())
        cb.max_inten_range = int((self.mx - self.mn).max().item())
        cb.max_bbox_side = int(max(self.h_bw.max(), self.h_bh.max()))
        self.cb = cb

    def host_rows(self, idx):
        """HostBatch of the ROIs `idx` (for the oracle)."""
        from nyxus_amd import _abi
        idx = np.asarray(idx, np.int64)
        offs = self.h_off.astype(np.int64)
        inten = self.inten.cpu().numpy().view(np.uint32)
        segs = [np.arange(offs[i], offs[i + 1]) for i in idx]
        cat = np.concatenate(segs)
        no = np.concatenate([[0], np.cumsum([len(s_) for s_ in segs])]).astype(np.uint64)
        mn, mx = self.mn.cpu().numpy().view(np.uint32), self.mx.cpu().numpy().view(np.uint32)
        return _abi.HostBatch(np.asarray(idx + 1, np.uint32), no, self.h_x[cat], self.h_y[cat], inten[cat], self.h_bw[idx], self.h_bh[idx], mn[idx], mx[idx])


def time_call(ctx, batch, mask, s, out, reps=3):
    import torch
    nc = out.shape[1]
    ctx.featurize_device_async(batch.cb, mask, s, out.data_ptr(), nc)
    torch.cuda.synchronize()
    ctx.sync()
    c0 = time.perf_counter()
    for _ in range(reps):
        ctx.featurize_device_async(batch.cb, mask, s, out.data_ptr(), nc)
    torch.cuda.synchronize()
    ctx.sync()
    return (time.perf_counter() - c0) / reps


SWEEP_RADII = (4, 9, 18, 30, 51, 102)          # disks of 49, 253, 1009, 2821, 8171, 32697 pixels


def size_sweep(ctx, dev, mask, s, px_budget=120_000_000, max_rois=196_000, radii=None):
    """ns per ROI of homogeneous batches, one per radius."""
    import torch
    rows = []
    for r in (radii or SWEEP_RADII):
        n1 = len(ellipse_cloud(r, r)[0])
        n_roi = int(max(256, min(max_rois, px_budget // n1)))
        b = DeviceBatch([(r, r)] * n_roi, dev, seed=100 + r)
        out = torch.empty((n_roi, ctx.n_columns(mask, s)), dtype=torch.float64, device=dev)
        dt = time_call(ctx, b, mask, s, out)
        rows.append({"n_px": n1, "box": 2 * r + 1, "rois": n_roi, "ns_per_roi": 1e9 * dt / n_roi, "rois_per_s": n_roi / dt,
                     "GBps": (8.0 * b.n_px + 8.0 * n_roi * out.shape[1]) / dt / 1e9})
        del b, out
    return rows


def mixed_shapes(n_roi=40_000, seed=3):
    """Log-normal radii clipped to 4..150 (median 12) + 1 % ellipses whose boxes are 300..400 px on a side."""
    rng = np.random.default_rng(seed)
    r = np.clip(np.rint(np.exp(rng.normal(np.log(12.0), 0.6, n_roi))), 4, 150).astype(int)
    shapes = [(int(v), int(v)) for v in r]
    for k in rng.choice(n_roi, n_roi // 100, replace=False):
        shapes[k] = (int(rng.integers(150, 200)), int(rng.integers(150, 200)))
    return shapes


def mixed_sizes(ctx, dev, mask, s, n_roi=40_000, check=None):
    """One call over the mixed batch; per size class: ROIs, share of pixels, and -- when the library reports it -- the class's time."""
    import torch
    shapes = mixed_shapes(n_roi)
    b = DeviceBatch(shapes, dev, seed=5)
    out = torch.empty((b.n_roi, ctx.n_columns(mask, s)), dtype=torch.float64, device=dev)
    dt = time_call(ctx, b, mask, s, out)
    rec = {"rois": b.n_roi, "pixels": b.n_px, "ms_per_call": 1e3 * dt, "rois_per_s": b.n_roi / dt,
           "GBps": (8.0 * b.n_px + 8.0 * b.n_roi * out.shape[1]) / dt / 1e9,
           "what": "log-normal radii 4..150 (median 12) + 1 % ellipses with 300..400-px boxes, one call"}
    n = b.n_px_roi
    edges = [0, 256, 1024, 4096, 16384, 65536, 1 << 30]
    rec["histogram"] = [{"px_le": e1, "rois": int(((n > e0) & (n <= e1)).sum()), "pixel_share": float(n[(n > e0) & (n <= e1)].sum() / n.sum())}
                        for e0, e1 in zip(edges[:-1], edges[1:])]
    if hasattr(ctx, "launch_report"):
        ctx.timing(True, groups=True)
        ctx.featurize_device_async(b.cb, mask, s, out.data_ptr(), out.shape[1])
        torch.cuda.synchronize()
        rec["classes"] = ctx.launch_report()
        ctx.timing(False)
    if check is not None:
        rec["parity_check"] = check(b, out)
    return rec, b, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--families", type=int, default=3)
    ap.add_argument("--gray-depth", type=int, default=8)
    ap.add_argument("--rois", type=int, default=40_000)
    ap.add_argument("--no-sweep", action="store_true")
    a = ap.parse_args()
    import torch
    from nyxus_amd import _abi, _lib
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = _lib.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    s = _abi.default_settings(a.gray_depth)
    rec = {}
    if not a.no_sweep:
        rec["size_sweep"] = size_sweep(ctx, dev, a.families, s)
    rec["mixed_sizes"], _, _ = mixed_sizes(ctx, dev, a.families, s, a.rois)
    print(json.dumps(rec))
    ctx.close()


if __name__ == "__main__":
    main()
