"""Per-call latency of the host-buffer entry points on small inputs (what an interactive Nyxus.featurize() call pays)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from nyxus_amd import _abi, _lib
from tests import synth

ctx = _lib.Context(0)
s = _abi.default_settings(8)
for n_tiles in (1, 4):
    b = synth.tile_batch(n_tiles, irregular=False, size=1024) if hasattr(synth, "tile_batch") else None
    for mask, name in ((3, "INT+GLCM"), (_abi.FAM_NORTH_STAR, "north-star 7")):
        for _ in range(3):
            ctx.featurize_host(b, mask, s)
        t0 = time.perf_counter()
        K = 20
        for _ in range(K):
            ctx.featurize_host(b, mask, s)
        dt = (time.perf_counter() - t0) / K
        print(f"batch path  {n_tiles} tile(s) x {b.n_roi // n_tiles} ROIs  {name:14s} {1e3 * dt:8.3f} ms per call")
