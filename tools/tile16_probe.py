"""Device tile path on full-range 16-bit tiles (u16 intensities, u8 labels): ms per call.  python tools/tile16_probe.py [tiles]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nyxus_amd import _abi, _lib
from tests import synth

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
ctx = _lib.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
s = _abi.default_settings(8)
mask = 3
lab = torch.from_numpy(synth.disk_label_tile().astype(np.uint8)).to(dev).unsqueeze(0).repeat(nt, 1, 1).contiguous()
g = torch.Generator(device=dev); g.manual_seed(5)
inten = torch.randint(1, 65536, (nt, 1024, 1024), generator=g, device=dev, dtype=torch.int32).to(torch.int16).contiguous()
ncol = ctx.n_columns(mask, s)
cap = nt * 196
t_lab = torch.empty(cap, dtype=torch.int32, device=dev); t_idx = torch.empty(cap, dtype=torch.int32, device=dev)
t_out = torch.empty((cap, ncol), dtype=torch.float64, device=dev)
t = _abi.Tiles()
t.inten = inten.data_ptr(); t.label = lab.data_ptr(); t.inten_dtype = _abi.U16; t.label_dtype = _abi.U8
t.width = 1024; t.height = 1024; t.n_tiles = nt; t.memory = _abi.MEM_DEVICE; t.slide_mode = _abi.SLIDE_MONTAGE
n = C.c_uint64(0)
lib = _lib.load()


def call():
    rc = lib.nyxhip_featurize_tiles_v2(ctx._h, C.byref(t), mask, C.byref(s), t_lab.data_ptr(), t_idx.data_ptr(), cap, t_out.data_ptr(), ncol, C.byref(n))
    assert rc == 0, lib.nyxhip_last_error(ctx._h)


call(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    call()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print({"tiles": nt, "rois": int(n.value), "ms": round(1e3 * dt, 3), "M_rois_per_s": round(n.value / dt / 1e6, 2), "no_wide": os.environ.get("NYXHIP_NO_WIDE")})
