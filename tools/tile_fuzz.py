"""Tile path (device label scan + compaction + cloud assembly + reduce) against the batch path on the same ROIs, for random
label images: touching / nested-looking blobs, sparse label values, ROIs cut by the tile border, a ROI that takes the global
workspace, empty tiles; random family subsets.  Both run on the GPU; the clouds are assembled in the same (row-major) order, so
the tables must agree bit for bit."""
import sys
import numpy as np
sys.path.insert(0, ".")
from nyxus_amd import _abi, _lib



def run(ctx, seed=0, rounds=20, verbose=True):
    """Returns the number of rounds in which the tile path and the batch path differ in any bit (or in the row keys)."""
    rng = np.random.default_rng(seed)
    bad_total = 0
    for rnd in range(rounds):
        nt = int(rng.integers(1, 4))
        H = int(rng.choice([64, 128, 200, 512])); W = int(rng.choice([64, 96, 256, 500]))
        lab = np.zeros((nt, H, W), np.uint32)
        for t in range(nt):
            if rng.random() < 0.15:
                continue                                            # empty tile
            n_blobs = int(rng.integers(1, 60))
            values = rng.choice(np.arange(1, 5000), n_blobs, replace=False)
            if rng.random() < 0.4:                                  # arbitrary 32-bit label values
                values = np.unique(rng.integers(1, 2 ** 32 - 1, n_blobs, dtype=np.uint64)).astype(np.uint32)
            yy, xx = np.mgrid[0:H, 0:W]
            for v in values:
                cy, cx = rng.integers(-5, H + 5), rng.integers(-5, W + 5)
                ry, rx = rng.integers(1, max(2, H // 3)), rng.integers(1, max(2, W // 3))
                if rng.random() < 0.1:
                    ry, rx = H, W                                       # a giant ROI (later blobs carve holes into it)
                m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1
                if rng.random() < 0.3:
                    m &= rng.random((H, W)) > 0.2
                lab[t][m] = v
        inten = rng.integers(0 if rng.random() < 0.3 else 1, int(rng.choice([50, 4096, 65536])), (nt, H, W)).astype(np.uint32)
        mask = int(rng.integers(1, 4096))
        gd = int(rng.choice([8, 16, 64]))
        s = _abi.default_settings(gd)
        if lab.max() == 0:
            continue
        inten_in, lab_in = inten, lab
        if inten.max() < 65536 and rng.random() < 0.5:             # the tile path takes the narrower element types as they are
            inten_in = inten.astype(np.uint16)
        if lab.max() < 65536 and rng.random() < 0.5:
            lab_in = lab.astype(np.uint16)
        try:
            tiles, labels, T = ctx.featurize_tiles_host(inten_in, lab_in, mask, s, max_device_bytes=int(rng.choice([0, 0, 8 << 20])))
        except _lib.NyxHipError as e:
            print("round", rnd, "tile path error", str(e)[:100]); bad_total += 1; continue
        rois, keys = [], []
        for t in range(nt):
            for v in np.unique(lab[t]):
                if v == 0:
                    continue
                ys, xs = np.nonzero(lab[t] == v)                       # row-major, like the device assembly
                rois.append(dict(x=xs, y=ys, inten=inten[t][ys, xs]))
                keys.append((t, int(v)))
        b = _abi.batch_from_rois(rois)
        B = ctx.featurize_host(b, mask, s)
        ok_keys = [(int(a), int(c)) for a, c in zip(tiles, labels)] == keys
        names = _lib.column_names(mask, s)
        cov = [j for j, n in enumerate(names) if n == "COVERED_IMAGE_INTENSITY_RANGE"]   # the tile path carries the montage's slide extrema
        same = (T == B) | (np.isnan(T) & np.isnan(B))
        if cov:
            same[:, cov] = True
        nbad = int((~same).sum())
        if not ok_keys or nbad:
            bad_total += 1
            cols = sorted({names[c] for c in np.nonzero(~same)[1]})[:6] if ok_keys else []
            print("round", rnd, "tiles", nt, H, W, "rois", len(rois), "mask", mask, "keys ok", ok_keys, "cells differing", nbad, cols, flush=True)
    return bad_total


if __name__ == "__main__":
    print("done; rounds with differences:", run(_lib.Context(0), int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 20))
