#!/usr/bin/env python3
"""How close each column comes to its tolerance: max over rows of |got - want| / tolerance (tests/parity.py), for every family on
the batches the GPU suite uses.  A column above 1 fails; a column that never exceeds 1e-6 has a tolerance looser than it needs.
    python tools/parity_margin.py [--top 40]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    from nyxus_amd import _abi, _lib
    from oracle import pyoracle as po
    from tests import parity, synth
    ctx = _lib.Context(0)
    worst = {}
    batches = [("random r25", _abi.batch_from_rois(synth.random_rois(60, seed=9, rmax=25))),
               ("random r12", _abi.batch_from_rois(synth.random_rois(60, seed=3, rmax=12))),
               ("random r40", _abi.batch_from_rois(synth.random_rois(60, seed=5, rmax=40))),
               ("irregular tile", synth.tile_batch(2, irregular=True))]
    for gd in (8, 64):
        s = _abi.default_settings(gd)
        mask = _abi.FAM_ALL & ~_abi.FAM_GABOR
        names = _lib.column_names(mask, s)
        for tag, b in batches:
            G = ctx.featurize_host(b, mask, s)
            O = po.oracle_featurize(b, mask, s)
            atol = parity.moment_atol(b, O, names)
            for j, n in enumerate(names):
                tol = parity.tolerance_of(n, O[:, j], atol=atol)
                g, w = G[:, j], O[:, j]
                same = (g == w) | (np.isnan(g) & np.isnan(w))
                with np.errstate(divide="ignore", invalid="ignore"):
                    ratio = np.where(same, 0.0, np.abs(g - w) / tol)
                ratio = np.where(np.isnan(ratio), np.inf, ratio)
                k = int(np.argmax(ratio))
                if ratio[k] > worst.get(n, (0,))[0]:
                    med = float(np.nanmedian(np.abs(w)))
                    worst[n] = (float(ratio[k]), f"gd {gd} {tag} roi {k}: got {g[k]!r} want {w[k]!r} tol {tol[k]:.3g} median|want| {med:.3g}")
    rows = sorted(worst.items(), key=lambda kv: -kv[1][0])
    print(f"{sum(1 for _, v in rows if v[0] > 1)} columns above their tolerance; top {a.top}:")
    for n, (r, where) in rows[: a.top]:
        print(f"{r:10.3g}  {n:28s} {where}")
    ctx.close()


if __name__ == "__main__":
    main()
