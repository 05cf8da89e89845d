"""Fuzz the several-workgroups-per-ROI texture path (nyxus_amd/csrc/roi_large_tex.hip) against the oracle: random boxes beyond the LDS
classes (and size class 2) -- thin, tall, wide up to a few thousand columns, with holes, flat patches, zero intensities, constant
and near-constant content -- under random binning modes, grey depths and subsets of GLRLM / GLSZM / NGTDM, alone and beside
INTENSITY / GLCM.  What it walks: strips of 1 .. thousands of rows, runs and zones that cross many strips, the drift pipeline of the
zone sweep at every width, 8- and 16-bit planes, the IBSI level maps.   python tools/ltex_fuzz.py [seed] [rounds]"""
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from nyxus_amd import _abi, _lib
from oracle import pyoracle as po
from tests import parity


def random_roi(rng, hi):
    kind = rng.integers(0, 6)
    if kind == 0:
        w, h = int(rng.integers(130, 420)), int(rng.integers(130, 420))
    elif kind == 1:
        w, h = int(rng.integers(300, 3000)), int(rng.integers(3, 40))          # wide strip
    elif kind == 2:
        w, h = int(rng.integers(3, 40)), int(rng.integers(300, 3000))          # tall strip
    elif kind == 3:
        w, h = int(rng.integers(65, 129)), int(rng.integers(65, 129))          # size class 2
    elif kind == 4:
        w, h = int(rng.integers(500, 900)), int(rng.integers(400, 700))
    else:
        w, h = int(rng.integers(64, 70)), int(rng.integers(250, 600))          # box widths around one chunk
    yy, xx = np.mgrid[0:h, 0:w]
    shape = rng.integers(0, 3)
    if shape == 0:
        m = (((xx - w / 2) / (w / 2 + .5)) ** 2 + ((yy - h / 2) / (h / 2 + .5)) ** 2) <= 1
    elif shape == 1:
        m = np.ones((h, w), bool)
    else:
        m = rng.random((h, w)) > rng.uniform(0.02, 0.4)
    if rng.random() < 0.4:
        m &= rng.random((h, w)) > 0.03
    m[0, 0] = m[h - 1, w - 1] = True                                           # the box keeps its size
    y, x = np.nonzero(m)
    o = np.lexsort((y, x)) if rng.random() < 0.7 else np.lexsort((x, y))      # column-major (in-memory workflow) or row-major clouds
    x, y = x[o], y[o]
    content = rng.integers(0, 5)
    lo = 0 if rng.random() < 0.3 else 1
    if content == 0:
        v = rng.integers(lo, hi, len(x))
    elif content == 1:
        v = rng.integers(lo, hi, len(x)); flat = ((x // int(rng.integers(5, 60))) + (y // int(rng.integers(5, 60)))) % 3 == 0; v[flat] = int(rng.integers(1, hi))
    elif content == 2:
        v = np.full(len(x), int(rng.integers(1, hi)))                           # constant
    elif content == 3:
        v = (rng.integers(lo, 4, len(x)) * (hi // 4)).clip(0, hi - 1)           # four values: huge zones
    else:
        v = ((x + y) % hi).astype(np.int64); v[v == 0] = lo                     # ramps: long diagonal runs
    return dict(x=x, y=y, inten=v.astype(np.uint32))


def run(ctx, seed=0, rounds=20, seconds=600, verbose=True):
    rng = np.random.default_rng(seed)
    bad_total = 0
    t0 = time.time()
    for rnd in range(rounds):
        mode = rng.integers(0, 4)
        ibsi = mode == 3
        gd = int(rng.choice([3, 8, 16, 64, 100, 300])) if mode != 1 else -int(rng.choice([4, 16, 40]))
        s = _abi.default_settings(gd, ibsi)
        hi = int(rng.choice([60, 200, 255])) if ibsi else int(rng.choice([16, 256, 4096, 70000]))
        fams = [_abi.FAM_GLRLM, _abi.FAM_GLSZM, _abi.FAM_NGTDM]
        mask = 0
        while mask == 0:
            mask = sum(f for f in fams if rng.random() < 0.6)
        if rng.random() < 0.3:
            mask |= _abi.FAM_INTENSITY | _abi.FAM_GLCM
        rois = [random_roi(rng, hi) for _ in range(int(rng.integers(1, 5)))]
        b = _abi.batch_from_rois(rois)
        G = ctx.featurize_host(b, mask, s)
        O = po.oracle_featurize(b, mask, s)
        # (a variance column of a matrix whose runs all have one length is 0 against 4e-31 of rounding: with a handful of rows the
        #  column scale offers no floor -- differences below 1e-18 absolute are not findings)
        with np.errstate(invalid="ignore"):
            G = np.where(np.abs(G - O) < 1e-18, O, G)
        bad = parity.compare_tables(G, O, _lib.column_names(mask, s), batch=b)
        coop = any(r["cooperative"] & 2 for r in ctx.launch_report())
        if verbose:
            print("round", rnd, "mask", mask, "gd", gd, "ibsi", ibsi, "hi", hi, "boxes", [(int(r["x"].max()) + 1, int(r["y"].max()) + 1) for r in rois], "coop", coop,
                  "bad", len(bad), bad[:3], flush=True)
        bad_total += len(bad)
        if time.time() - t0 > seconds:
            break
    return bad_total


if __name__ == "__main__":
    print("done; mismatches:", run(_lib.Context(0), int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 20))
