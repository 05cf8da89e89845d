"""195 999 disks of radius argv[1] + ONE disk of radius 30: the stated extrema admit size class 1, so the smallest class is served by the
SCANNING form of roi_small_kernel (a filtered whole-batch launch).  argv[2] = family mask, argv[3] = grey depth.  ns per ROI."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, size_legs as sl
from nyxus_amd import _abi, _lib
r = int(sys.argv[1]) if len(sys.argv) > 1 else 4
fam = int(sys.argv[2]) if len(sys.argv) > 2 else 3
gd = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda", 0); ctx = _lib.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
s = _abi.default_settings(gd)
b = sl.DeviceBatch([(r, r)] * 195999 + [(30, 30)], dev, seed=3)
out = torch.empty((b.n_roi, ctx.n_columns(fam, s)), dtype=torch.float64, device=dev)
dt = sl.time_call(ctx, b, fam, s, out, reps=5)
print("mixed: radius", r, "families", fam, "grey depth", gd, "ns per ROI", round(1e9 * dt / b.n_roi, 2), [x["class"] for x in ctx.launch_report()])
ctx.close()
