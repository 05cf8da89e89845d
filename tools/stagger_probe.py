"""Two contexts on one GPU, each looping over HALF of a device-resident tile stack, the second one started `argv[2]` ms after the first: is the
HBM-bound label scan of one half hidden behind the VALU-bound reduce of the other when the two calls are out of phase?  (tools/two_ctx_probe.py
starts them in phase.)  python tools/stagger_probe.py [tiles] [delay_ms] [loops]"""
import ctypes as C, sys, threading, time
import numpy as np, torch
sys.path.insert(0, ".")
from nyxus_amd import _abi, _lib
from tests import synth

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
delay = float(sys.argv[2]) if len(sys.argv) > 2 else 1.1
loops = int(sys.argv[3]) if len(sys.argv) > 3 else 8
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

h = nt // 2
for k in range(2): run(ctxs[k], k * h, h, outs[k])
torch.cuda.synchronize()
# one context, whole stack
c0 = time.perf_counter()
for _ in range(loops // 2): n = run(ctxs[0], 0, nt, outs[0])
torch.cuda.synchronize()
dt1 = (time.perf_counter() - c0) / (loops // 2)
print(f"one context, {nt} tiles per call: {1e3 * dt1:.2f} ms = {n / dt1 / 1e6:.1f} M ROIs/s", flush=True)
tot = [0, 0]
def worker(k):
    if k: time.sleep(delay * 1e-3)
    for _ in range(loops): tot[k] += run(ctxs[k], k * h, h, outs[k])
c0 = time.perf_counter()
th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
for t in th: t.start()
for t in th: t.join()
torch.cuda.synchronize()
dt2 = time.perf_counter() - c0
print(f"two contexts, {h} tiles per call, second delayed {delay} ms, {loops} calls each: {1e3 * dt2:.2f} ms total = {sum(tot) / dt2 / 1e6:.1f} M ROIs/s", flush=True)
