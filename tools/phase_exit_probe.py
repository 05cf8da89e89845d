#!/usr/bin/env python3
"""Time of a homogeneous batch of disks through the library named by NYXHIP_LIB (an early-exit build: -DNYX_EXIT_AT=k ends
roi_features_kernel at STAMP(k); results are wrong by design) -- the differences between consecutive k are the phases of the kernel
at that ROI size, waiting included.  python tools/phase_exit_probe.py [--radius 4] [--rois 196000]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--radius", type=int, default=4)
    ap.add_argument("--rois", type=int, default=196000)
    a = ap.parse_args()
    import torch
    import size_legs as sl
    from nyxus_amd import _abi, _lib
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    s = _abi.default_settings(8)
    b = sl.DeviceBatch([(a.radius, a.radius)] * a.rois, dev, seed=3)
    out = torch.empty((b.n_roi, ctx.n_columns(3, s)), dtype=torch.float64, device=dev)
    dt = sl.time_call(ctx, b, 3, s, out, reps=5)
    print(json.dumps({"lib": os.path.basename(os.environ.get("NYXHIP_LIB", "libnyxhip.so")), "n_px": int(b.n_px_roi[0]), "ns_per_roi": round(1e9 * dt / b.n_roi, 2)}))
    ctx.close()


if __name__ == "__main__":
    main()
