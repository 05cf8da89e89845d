"""GLCM settings fuzz: angle subsets in any order, symmetric flag, offsets 1..4, grey depths around the split / 8-bit-plane
thresholds, boxes narrower and wider than a wave -- against the oracle (INTENSITY + GLCM and GLCM alone)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from nyxus_amd import _abi, _lib
from oracle import pyoracle as po
from tests import parity, synth

ctx = _lib.Context(0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
bad_total = 0
for rnd in range(rounds):
    mode = rng.integers(0, 3)
    ibsi = mode == 2
    gd = int(rng.choice([2, 3, 8, 15, 16, 17, 32, 64])) if mode != 1 else -int(rng.choice([3, 8, 16, 30]))
    s = _abi.default_settings(gd, ibsi)
    k = int(rng.integers(1, 5))
    angs = list(rng.permutation([0, 45, 90, 135])[:k])
    s.glcm_n_angles = k
    for i, a in enumerate(angs):
        s.glcm_angles[i] = int(a)
    s.glcm_offset = int(rng.integers(1, 5))
    s.glcm_symmetric = int(rng.random() < 0.5)
    rois = synth.random_rois(int(rng.integers(5, 60)), seed=int(rng.integers(0, 1 << 30)), rmax=int(rng.choice([6, 14, 30])))
    if rng.random() < 0.4:
        h, w = int(rng.integers(20, 90)), int(rng.integers(65, 140))
        ys, xs = np.nonzero(rng.random((h, w)) > 0.1)
        rois.append(dict(x=xs, y=ys, inten=rng.integers(0, 300, len(xs)).astype(np.uint32)))
    if ibsi:
        rois = [dict(r, inten=(np.asarray(r["inten"]) % int(rng.choice([5, 9, 40]))).astype(np.uint32)) for r in rois]
        rois = [r for r in rois if np.asarray(r["inten"]).max() > 0]
    b = _abi.batch_from_rois(rois)
    mask = int(rng.choice([2, 3]))
    try:
        G = ctx.featurize_host(b, mask, s)
    except _lib.NyxHipError as e:
        print("round", rnd, "error", str(e)[:100]); bad_total += 1; continue
    O = po.oracle_featurize(b, mask, s)
    bad = parity.compare_tables(G, O, _lib.column_names(mask, s))
    if bad:
        bad_total += 1
        print("round", rnd, "gd", gd, "ibsi", ibsi, "angles", angs, "off", s.glcm_offset, "sym", s.glcm_symmetric, "mask", mask, len(bad), bad[:2], flush=True)
print("done; rounds with mismatches:", bad_total)
