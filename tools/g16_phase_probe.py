#!/usr/bin/env python3
"""ns per ROI of GLCM alone at grey depth 64 on homogeneous batches of disks (49 / 253 / 2821 px) through the library named by
NYXHIP_LIB -- with the early-exit builds of tools/g16_exit_libs.sh the differences of consecutive k are the phases of
glcm_features_wave64_v2 (results of those builds are wrong by design)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import torch
    import size_legs as sl
    from nyxus_amd import _abi, _lib
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    s = _abi.default_settings(64)
    res = {}
    for r, n in ((3, 196000), (4, 196000), (9, 196000), (30, 98000)):
        b = sl.DeviceBatch([(r, r)] * n, dev, seed=3)
        out = torch.empty((b.n_roi, ctx.n_columns(2, s)), dtype=torch.float64, device=dev)
        dt = sl.time_call(ctx, b, 2, s, out, reps=5)
        res[int(b.n_px_roi[0])] = round(1e9 * dt / b.n_roi, 2)
    print(json.dumps({"lib": os.path.basename(os.environ.get("NYXHIP_LIB", "libnyxhip.so")), "ns_per_roi": res}))
    ctx.close()


if __name__ == "__main__":
    main()
