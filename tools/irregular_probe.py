"""SURVEY 8(d)'s second synthetic set through the device tile path: the 14 x 14 grid with per-ROI radius in [8, 36) and 10 % concave
ROIs (load imbalance, background inside the boxes).  Prints ROIs/s for the metric families and for BASELINE.json configs[3]'s five."""
import ctypes as C, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from nyxus_amd import _abi, _lib
from tests import synth

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
nvar = 8                                                   # distinct irregular label tiles, cycled
lab = torch.from_numpy(np.stack([synth.disk_label_tile(irregular=True, seed=k) for k in range(nvar)]).astype(np.int32)).to(dev)
labs = lab.repeat((nt + nvar - 1) // nvar, 1, 1)[:nt].contiguous()
tin = torch.randint(1, 4096, (nt, 1024, 1024), generator=g, device=dev, dtype=torch.int32)
lib = _lib.load()
ctx = _lib.Context(0)
for name, mask, gd in (("INTENSITY + GLCM, grey depth 8", 3, 8), ("+ GLRLM + GLSZM + NGTDM, grey depth 8", 31, 8), ("INTENSITY + GLCM, grey depth 64", 3, 64)):
    s = _abi.default_settings(gd)
    ncol = ctx.n_columns(mask, s)
    cap = nt * 196
    o_lab = torch.empty(cap, dtype=torch.int32, device=dev); o_idx = torch.empty(cap, dtype=torch.int32, device=dev)
    o_tab = torch.empty((cap, ncol), dtype=torch.float64, device=dev)
    nroi = C.c_uint64(0)

    def run():
        rc = lib.nyxhip_featurize_tiles(ctx._h, tin.data_ptr(), labs.data_ptr(), 1024, 1024, nt, _abi.MEM_DEVICE, 196, mask,
                                        C.byref(s), o_lab.data_ptr(), o_idx.data_ptr(), cap, o_tab.data_ptr(), ncol, C.byref(nroi))
        assert rc == 0, lib.nyxhip_last_error(ctx._h)
    run(); torch.cuda.synchronize()
    c0 = time.perf_counter()
    for _ in range(3): run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - c0) / 3
    print(f"irregular set, {name}: {nroi.value} ROIs in {1e3 * dt:.2f} ms = {nroi.value / dt / 1e6:.1f} M ROIs/s", flush=True)
