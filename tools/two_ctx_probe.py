"""Device-resident tile stack through ONE context vs split over TWO contexts on the same GPU (two host threads): does the
HBM-bound label scan of one half overlap the VALU-bound reduce of the other?"""
import ctypes as C, sys, threading, time
import numpy as np, torch
sys.path.insert(0, ".")
from nyxus_amd import _abi, _lib
from tests import synth

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
lab1 = torch.from_numpy(synth.disk_label_tile().astype(np.int32)).to(dev)
labs = lab1.unsqueeze(0).repeat(nt, 1, 1).contiguous()
tin = torch.randint(1, 4096, (nt, 1024, 1024), generator=g, device=dev, dtype=torch.int32)
s = _abi.default_settings(8)
mask = 3
lib = _lib.load()
ctxs = [_lib.Context(0), _lib.Context(0)]
ncol = ctxs[0].n_columns(mask, s)
cap = nt * 196
outs = [(torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.int32, device=dev), torch.empty((cap, ncol), dtype=torch.float64, device=dev)) for _ in range(2)]

def run(ctx, t0, n, out):
    nroi = C.c_uint64(0)
    rc = lib.nyxhip_featurize_tiles(ctx._h, tin[t0:t0 + n].data_ptr(), labs[t0:t0 + n].data_ptr(), 1024, 1024, n, _abi.MEM_DEVICE, 196, mask,
                                    C.byref(s), out[0].data_ptr(), out[1].data_ptr(), cap, out[2].data_ptr(), ncol, C.byref(nroi))
    assert rc == 0, lib.nyxhip_last_error(ctx._h)
    return nroi.value

def one():
    return run(ctxs[0], 0, nt, outs[0])

def two():
    h = nt // 2
    r = [0, 0]
    th = [threading.Thread(target=lambda k=k: r.__setitem__(k, run(ctxs[k], k * h, h if k == 0 else nt - h, outs[k]))) for k in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    return sum(r)

for name, f in (("one context", one), ("two contexts", two), ("one context", one), ("two contexts", two)):
    f(); torch.cuda.synchronize()
    c0 = time.perf_counter()
    for _ in range(3): n = f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - c0) / 3
    print(f"{name}: {n} ROIs in {1e3 * dt:.2f} ms = {n / dt / 1e6:.1f} M ROIs/s", flush=True)
