"""Random FAMILY SUBSETS x random settings x batches that mix LDS-sized, wide (> 64 px) and global-workspace ROIs, against the
oracle.  The carve-outs, register tiers and launch order of the kernels depend on the family set and on the batch extrema, so
this walks combinations the per-family tests do not."""
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from nyxus_amd import _abi, _lib
from oracle import pyoracle as po
from tests import parity

SOFT = ("WNCM", "WHU", "WT_NORM", "WEIGHTED_HU", "IMOM_WCM", "WEIGHTED_CENTRAL")


def run(ctx, seed=0, rounds=30, seconds=420, verbose=True):
    """Returns the number of hard mismatches over `rounds` random (family subset, settings, batch) draws."""
    rng = np.random.default_rng(seed)
    print_ = print if verbose else (lambda *a, **k: None)


    def blob(h, w):
        yy, xx = np.mgrid[0:h, 0:w]
        m = (((xx - w / 2) / (w / 2 + .5)) ** 2 + ((yy - h / 2) / (h / 2 + .5)) ** 2) <= 1
        k = rng.integers(0, 3)
        if k == 1:
            m = np.ones((h, w), bool)
        elif k == 2:
            m &= rng.random((h, w)) > 0.15
        if not m.any():
            m[0, 0] = True
        ys, xs = np.nonzero(m)
        return xs - xs.min(), ys - ys.min()


    total_hard = 0
    t0 = time.time()
    for rnd in range(rounds):
        mode = rng.integers(0, 4)
        ibsi = mode == 3
        gd = int(rng.choice([3, 8, 16, 20, 64, 200])) if mode != 1 else -int(rng.choice([4, 16, 40]))
        s = _abi.default_settings(gd, ibsi)
        s.glcm_n_angles = int(rng.integers(1, 5))
        mask = int(rng.integers(1, 4096))
        if gd < 0 and not ibsi:
            mask &= ~(_abi.FAM_GLDZM | _abi.FAM_NGLDM)
        big = rng.random() < 0.35                      # a ROI for the global workspace
        wide = rng.random() < 0.5                      # ROIs wider than one wave
        if big or wide:
            mask &= ~_abi.FAM_GABOR if rng.random() < 0.7 else mask     # the CPU oracle's Gabor is slow on large boxes
        if mask == 0:
            mask = _abi.FAM_INTENSITY
        dims = [tuple(rng.integers(1, 50, 2)) for _ in range(int(rng.integers(3, 40)))]
        if wide:
            dims += [tuple(rng.integers(65, 160, 2)) for _ in range(int(rng.integers(1, 4)))]
        if big:
            dims += [(int(rng.integers(200, 420)), int(rng.integers(200, 420)))]
        rng.shuffle(dims)
        rois = []
        for (h, w) in dims:
            xs, ys = blob(int(h), int(w))
            n = len(xs)
            if ibsi:
                v = rng.integers(0 if rng.random() < .3 else 1, int(rng.choice([3, 7, 20, 60])), n)
                if v.max() == 0:
                    v[0] = 1
            else:
                d = rng.integers(0, 4)
                v = (rng.integers(1, 4096, n) if d == 0 else rng.integers(0, 12, n) if d == 1
                     else rng.integers(30000, 65536, n) if d == 2 else (rng.normal(1000, 40, n)).clip(0).astype(np.int64))
            rois.append(dict(x=xs, y=ys, inten=np.asarray(v).astype(np.uint32)))
        b = _abi.batch_from_rois(rois)
        try:
            G = ctx.featurize_host(b, mask, s)
        except _lib.NyxHipError as e:
            print_("round", rnd, "mask", mask, "gd", gd, "ibsi", ibsi, "-> error", str(e)[:100], flush=True)
            continue
        O = po.oracle_featurize(b, mask, s)
        bad = parity.compare_tables(G, O, _lib.column_names(mask, s), atol=parity.moment_atol(b))
        hard = [x for x in bad if not any(t in x for t in SOFT)]
        total_hard += len(hard)
        print_("round", rnd, "mask", mask, "gd", gd, "ibsi", ibsi, "rois", len(rois), "big", big, "wide", wide, "hard", len(hard), hard[:2], flush=True)
        if time.time() - t0 > seconds:
            break
    return total_hard


if __name__ == "__main__":
    print("done; hard mismatches:", run(_lib.Context(0), int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 30))
