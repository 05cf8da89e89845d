"""Device-resident stack of the benchmark's tiles through nyxhip_featurize_tiles_v2 (INTENSITY + GLCM, grey depth 8 / argv[2]): ms per call."""
import ctypes as C, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from nyxus_amd import _abi, _lib
from tests import synth

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
gd = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
lab1 = torch.from_numpy(synth.disk_label_tile().astype(np.int32)).to(dev)
labs = lab1.unsqueeze(0).repeat(nt, 1, 1).contiguous()
tin = torch.randint(1, 4096, (nt, 1024, 1024), generator=g, device=dev, dtype=torch.int32)
s = _abi.default_settings(gd)
mask = 3
lib = _lib.load()
ctx = _lib.Context(0)
ncol = ctx.n_columns(mask, s)
cap = nt * 196
out = (torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.int32, device=dev), torch.empty((cap, ncol), dtype=torch.float64, device=dev))

def run():
    nroi = C.c_uint64(0)
    rc = lib.nyxhip_featurize_tiles(ctx._h, tin.data_ptr(), labs.data_ptr(), 1024, 1024, nt, _abi.MEM_DEVICE, 196, mask,
                                    C.byref(s), out[0].data_ptr(), out[1].data_ptr(), cap, out[2].data_ptr(), ncol, C.byref(nroi))
    assert rc == 0, lib.nyxhip_last_error(ctx._h)
    return nroi.value

run(); torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    c0 = time.perf_counter()
    for _ in range(3): n = run()
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - c0) / 3)
print(f"tiles {nt} gd {gd}: {n} ROIs in {1e3 * best:.3f} ms = {n / best / 1e6:.2f} M ROIs/s  checksum {float(out[2][:n].nan_to_num().sum()):.6e}", flush=True)
