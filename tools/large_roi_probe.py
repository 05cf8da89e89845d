#!/usr/bin/env python3
"""Where a large ROI spends its time: a batch of N equal ellipses (semi-axes a, b) through one call, per size class times from
nyxhip_launch_report; with the stamped diagnostic build (make -C nyxus_amd/csrc stamp; NYXHIP_LIB=nyxus_amd/libnyxhip_stamp.so
NYXHIP_STAMPS=1) the per-phase cycle shares of roi_features_kernel are printed when the context closes.
    python tools/large_roi_probe.py --a 180 --b 170 --n 256 --families 3
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--a", type=int, default=180)
    ap.add_argument("--b", type=int, default=170)
    ap.add_argument("--n", type=int, default=256)
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
    b = sl.DeviceBatch([(a.a, a.b)] * a.n, dev, seed=9, hi=a.hi)
    out = torch.empty((b.n_roi, ctx.n_columns(a.families, s)), dtype=torch.float64, device=dev)
    dt = sl.time_call(ctx, b, a.families, s, out, reps=2)
    ctx.timing(True, groups=True)
    ctx.featurize_device_async(b.cb, a.families, s, out.data_ptr(), out.shape[1])
    torch.cuda.synchronize()
    rep = ctx.launch_report()
    ctx.timing(False)
    print(json.dumps({"n_px": int(b.n_px_roi[0]), "box": [2 * a.a + 1, 2 * a.b + 1], "rois": a.n, "ms_per_call": 1e3 * dt,
                      "us_per_roi": 1e6 * dt / a.n, "GBps": 8.0 * b.n_px / dt / 1e9, "classes": rep}))
    ctx.close()


if __name__ == "__main__":
    main()
