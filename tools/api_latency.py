"""Wall time of the Python face (Nyxus.featurize -> pandas DataFrame) next to the C-ABI call underneath it."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import nyxus_amd
from tests import synth

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 32
inten, seg = synth.tile_stack(nt) if hasattr(synth, "tile_stack") else (None, None)
if inten is None:
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:1024, 0:1024]
    seg1 = np.zeros((1024, 1024), np.uint32)
    k = 1
    for cy in range(36, 1024, 73):
        for cx in range(36, 1024, 73):
            seg1[(yy - cy) ** 2 + (xx - cx) ** 2 <= 900] = k; k += 1
    seg = np.repeat(seg1[None], nt, 0)
    inten = rng.integers(1, 4096, seg.shape).astype(np.uint32)
nyx = nyxus_amd.Nyxus(["*ALL_INTENSITY*", "*ALL_GLCM*"], coarse_gray_depth=8)
for _ in range(2):
    df = nyx.featurize(inten, seg)
t0 = time.perf_counter()
K = 5
for _ in range(K):
    df = nyx.featurize(inten, seg)
dt = (time.perf_counter() - t0) / K
print(f"Nyxus.featurize: {nt} tiles, {len(df)} rows x {df.shape[1]} cols: {1e3 * dt:.1f} ms per call ({len(df) / dt:.0f} ROIs/s)")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); nyx.featurize(inten, seg); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
