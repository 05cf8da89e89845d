"""Homogeneous batch of 196 000 disks of radius argv[1] (default 4: 49 px) through INTENSITY + GLCM (argv[2] = family mask): ns per ROI."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, size_legs as sl
from nyxus_amd import _abi, _lib
r = int(sys.argv[1]) if len(sys.argv) > 1 else 4
fam = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0); ctx = _lib.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
gd = int(sys.argv[3]) if len(sys.argv) > 3 else 8
s = _abi.default_settings(gd)
b = sl.DeviceBatch([(r, r)] * 196000, dev, seed=3)
out = torch.empty((b.n_roi, ctx.n_columns(fam, s)), dtype=torch.float64, device=dev)
dt = sl.time_call(ctx, b, fam, s, out, reps=5)
print("radius", r, "px", int(b.n_px_roi[0]), "families", fam, "grey depth", gd, "ns per ROI", round(1e9 * dt / b.n_roi, 2))
ctx.close()
