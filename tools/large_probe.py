#!/usr/bin/env python3
"""Times one homogeneous batch of large ROIs through the C ABI (large-ROI path experiments):
python tools/large_probe.py [--radius 102] [--rois 3671] [--families 3] [--hi 4096]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--radius", type=int, default=102)
    ap.add_argument("--rois", type=int, default=3671)
    ap.add_argument("--families", type=int, default=3)
    ap.add_argument("--gray-depth", type=int, default=8)
    ap.add_argument("--hi", type=int, default=4096)
    a = ap.parse_args()
    import torch
    import size_legs as sl
    from nyxus_amd import _abi, _lib
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    s = _abi.default_settings(a.gray_depth)
    b = sl.DeviceBatch([(a.radius, a.radius)] * a.rois, dev, seed=3, hi=a.hi)
    out = torch.empty((b.n_roi, ctx.n_columns(a.families, s)), dtype=torch.float64, device=dev)
    dt = sl.time_call(ctx, b, a.families, s, out)
    print(json.dumps({"radius": a.radius, "rois": b.n_roi, "n_px": int(b.n_px_roi[0]), "ms": 1e3 * dt, "ns_per_roi": 1e9 * dt / b.n_roi, "dbg": os.environ.get("NYXHIP_LARGE_DBG")}))
    ctx.close()


if __name__ == "__main__":
    main()
