"""Gabor: exact kernel (default) vs fused-multiply-add kernel (NYXHIP_GABOR_FUSED=1), both on the GPU.

    python tools/gabor_fuzz.py save <file.npy> [seed] [n_rois]   -> Gabor table of a random batch under the current environment
    python tools/gabor_fuzz.py cmp <a.npy> <b.npy>               -> rows / cells that differ

The two `save` runs are separate processes because the kernel choice is read once per process.  Shapes: random blobs with
bounding boxes of 3..70 px, smooth + noisy + flat intensity fields (flat fields make thousands of pixels share one energy,
the worst case for a threshold count)."""
import sys
import numpy as np
sys.path.insert(0, ".")


MAX_SIDE = int(__import__("os").environ.get("GABOR_FUZZ_MAX_SIDE", "70"))   # 70: small + medium classes; up to ~180 still takes the tiled kernel


def batch(seed, n_rois):
    rng = np.random.default_rng(seed)
    rois = []
    for k in range(n_rois):
        h, w = rng.integers(3, MAX_SIDE + 1, 2)
        yy, xx = np.mgrid[0:h, 0:w]
        m = (((xx - w / 2) / (w / 2 + .5)) ** 2 + ((yy - h / 2) / (h / 2 + .5)) ** 2) <= 1
        kind = rng.integers(0, 5)
        if kind == 1:
            m = np.ones((h, w), bool)
        elif kind == 2:
            m &= rng.random((h, w)) > 0.1
        if not m.any():
            m[0, 0] = True
        ys, xs = np.nonzero(m)
        xs = xs - xs.min(); ys = ys - ys.min()
        n = len(xs)
        field = rng.integers(0, 5)
        if field == 0:
            v = rng.integers(1, 4096, n)
        elif field == 1:
            v = (1000 + 400 * np.sin(xs / 3.0) * np.cos(ys / 5.0) + rng.normal(0, 20, n)).clip(1)
        elif field == 2:
            v = np.where((xs // 4 + ys // 4) % 2 == 0, 200, 3000)     # checkerboard of flat blocks
        elif field == 3:
            v = np.full(n, 777); v[rng.integers(0, n)] = 778           # flat but for one pixel
        else:
            v = rng.integers(0, 2 ** 16, n)
        rois.append(dict(x=xs, y=ys, inten=np.asarray(v).astype(np.uint32)))
    return rois


if sys.argv[1] == "save":
    from nyxus_amd import _abi, _lib
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    n_rois = int(sys.argv[4]) if len(sys.argv) > 4 else 20000
    ctx = _lib.Context(0)
    s = _abi.default_settings(8, False)
    G = ctx.featurize_host(_abi.batch_from_rois(batch(seed, n_rois)), _abi.FAM_GABOR, s)
    np.save(sys.argv[2], G)
    print("saved", G.shape, "nan rows", int(np.isnan(G).any(axis=1).sum()))
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    print("rows", a.shape[0], "rows differing", int((~same).any(axis=1).sum()), "cells differing", int((~same).sum()),
          "max rel diff", float(np.nanmax(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if (~same).any() else 0.0)
    if (~same).any() and len(sys.argv) > 5:
        # which intensity fields do the differing rows come from?  (replays the generator's draws)
        seed, n_rois = int(sys.argv[4]), int(sys.argv[5])
        rng = np.random.default_rng(seed)
        kinds = []
        for k in range(n_rois):
            h, w = rng.integers(3, 71, 2)
            yy, xx = np.mgrid[0:h, 0:w]
            m = (((xx - w / 2) / (w / 2 + .5)) ** 2 + ((yy - h / 2) / (h / 2 + .5)) ** 2) <= 1
            kind = rng.integers(0, 5)
            if kind == 1:
                m = np.ones((h, w), bool)
            elif kind == 2:
                m &= rng.random((h, w)) > 0.1
            if not m.any():
                m[0, 0] = True
            n = int(m.sum())
            field = rng.integers(0, 5)
            if field == 0: rng.integers(1, 4096, n)
            elif field == 1: rng.normal(0, 20, n)
            elif field == 3: rng.integers(0, n)
            elif field == 4: rng.integers(0, 2 ** 16, n)
            kinds.append(int(field))
        kinds = np.array(kinds)
        bad = (~same).any(axis=1)
        print("differing rows by intensity field:", {int(f): int((bad & (kinds == f)).sum()) for f in range(5)},
              "of", {int(f): int((kinds == f).sum()) for f in range(5)})
