"""Gabor bank fuzz against the oracle: 1..16 filters with random (f0, theta), kernel sizes 5..24 (16 takes the register-tiled
kernel, the rest the generic one), thresholds, gamma / sig2lam; small and mid-size ROIs.  Count ratios must match exactly."""
import sys
import numpy as np
sys.path.insert(0, ".")
from nyxus_amd import _abi, _lib
from oracle import pyoracle as po
from tests import synth

ctx = _lib.Context(0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(seed)
bad_total = 0
for rnd in range(rounds):
    s = _abi.default_settings(8)
    nf = int(rng.integers(1, 17))
    s.gabor_n_filters = nf
    for i in range(nf):
        s.gabor_f0[i] = float(rng.choice([1.0, 4.0, 16.0, 32.0, 64.0, 100.0]) * rng.uniform(0.8, 1.2))
        s.gabor_theta[i] = float(rng.uniform(0, np.pi))
    s.gabor_kersize = int(rng.choice([16, 16, 16, 5, 9, 20, 24]))
    s.gabor_graythr = float(rng.choice([0.025, 0.1, 0.5, 0.0]))
    s.gabor_gamma = float(rng.choice([0.1, 0.5, 1.0]))
    s.gabor_sig2lam = float(rng.choice([0.8, 0.4, 1.5]))
    s.gabor_f0lp = float(rng.choice([0.1, 0.05, 1.0]))
    rmax = int(rng.choice([6, 12, 25]))
    rois = synth.random_rois(int(rng.integers(3, 25)), seed=int(rng.integers(0, 1 << 30)), rmax=rmax)
    b = _abi.batch_from_rois(rois)
    try:
        G = ctx.featurize_host(b, _abi.FAM_GABOR, s)
    except _lib.NyxHipError as e:
        print("round", rnd, "error", str(e)[:100]); bad_total += 1; continue
    O = po.oracle_featurize(b, _abi.FAM_GABOR, s)
    same = (G == O) | (np.isnan(G) & np.isnan(O))
    if not same.all():
        bad_total += 1
        print("round", rnd, "nf", nf, "n", s.gabor_kersize, "thr", s.gabor_graythr, "cells differing", int((~same).sum()),
              "max abs", float(np.nanmax(np.abs(G - O))), flush=True)
print("done; rounds with differences:", bad_total)
