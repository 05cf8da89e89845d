"""Fuzz the contour + moments kernels against the oracle on adversarial masks: noise at several densities, thin lines,
checkerboards, rings, specks.  Prints the first mismatches."""
import sys
import numpy as np
sys.path.insert(0, ".")
from nyxus_amd import _abi, _lib
from oracle import pyoracle as po
from tests import parity

ctx = _lib.Context(0)
s = _abi.default_settings(8)
mask = _abi.FAM_SMOMS | _abi.FAM_IMOMS
names = _lib.column_names(mask, s)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_bad = 0
for rnd in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    rois = []
    for k in range(200):
        h, w = rng.integers(1, 40, 2)
        kind = rng.integers(0, 6)
        if kind == 0:
            m = rng.random((h, w)) < rng.choice([0.1, 0.3, 0.5, 0.7, 0.9])
        elif kind == 1:
            yy, xx = np.mgrid[0:h, 0:w]; m = ((xx + yy) % 2 == 0)
        elif kind == 2:
            m = np.zeros((h, w), bool); m[rng.integers(0, h), :] = True; m[:, rng.integers(0, w)] = True
        elif kind == 3:
            yy, xx = np.mgrid[0:h, 0:w]; r = np.hypot(xx - w / 2, yy - h / 2); m = (r < min(h, w) / 2) & (r > min(h, w) / 4)
        elif kind == 4:
            m = rng.random((h, w)) < 0.6; m[1:-1:2, :] = False
        else:
            yy, xx = np.mgrid[0:h, 0:w]; m = (np.abs(xx - yy) <= 1) | (rng.random((h, w)) < 0.05)
        if not m.any():
            m[0, 0] = True
        ys, xs = np.nonzero(m)
        xs = xs - xs.min(); ys = ys - ys.min()
        rois.append(dict(x=xs, y=ys, inten=rng.integers(0, 500, len(xs)).astype(np.uint32)))
    b = _abi.batch_from_rois(rois)
    G = ctx.featurize_host(b, mask, s)
    O = po.oracle_featurize(b, mask, s)
    bad = parity.compare_tables(G, O, names, batch=b)          # (the floors of the suite: the weighted sets derive theirs from the table -- a weighted mass that nearly cancels puts its origin far outside the box)
    if bad:
        n_bad += len(bad)
        print("round", rnd, len(bad), bad[:4])
print("done, mismatches:", n_bad)
