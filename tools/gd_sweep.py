import sys, os, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from nyxus_amd import _abi, _lib
import size_legs as sl
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = _lib.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for gd in (8, 64):
    s = _abi.default_settings(gd)
    for fam in (3, 2, 1):
        rows = sl.size_sweep(ctx, dev, fam, s)
        print("gd", gd, "families", fam, {r["n_px"]: round(r["ns_per_roi"], 2) for r in rows})
ctx.close()
