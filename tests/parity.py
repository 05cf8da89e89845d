"""Comparison policy shared by the parity tests (north_star):
bit-exact for label/count features, <= 1e-5 relative otherwise, with an absolute floor
of 1e-9 x (feature scale) for features that cross zero through cancellation."""
from __future__ import annotations

import numpy as np

REL_TOL = 1e-5
# integer-exact columns: sums / order statistics of integers and integer-valued ratios
EXACT_COLUMNS = {"MIN", "MAX", "RANGE", "MODE", "MEDIAN", "INTEGRATED_INTENSITY", "ENERGY", "MEAN",
                 "ROOT_MEAN_SQUARED", "P01", "P10", "P25", "P75", "P90", "P99", "INTERQUARTILE_RANGE", "QCOD",
                 "UNIFORMITY_PIU", "COVERED_IMAGE_INTENSITY_RANGE", "ROBUST_MEAN"}


def compare_tables(got: np.ndarray, want: np.ndarray, names, rel=REL_TOL, exact=EXACT_COLUMNS):
    """Returns a list of human-readable mismatches (empty = parity)."""
    assert got.shape == want.shape, (got.shape, want.shape)
    bad = []
    for j, name in enumerate(names):
        g, w = got[:, j], want[:, j]
        both_nan = np.isnan(g) & np.isnan(w)
        same_inf = np.isinf(g) & np.isinf(w) & (np.sign(g) == np.sign(w))
        ok = both_nan | same_inf | (g == w)
        base = name
        for suf in ("_0", "_45", "_90", "_135"):
            if base.endswith(suf):
                base = base[: -len(suf)]
        if base not in exact:
            scale = np.nanmax(np.abs(np.where(np.isfinite(w), w, 0.0))) if len(w) else 0.0
            with np.errstate(invalid="ignore"):
                ok |= np.abs(g - w) <= rel * np.abs(w) + 1e-9 * max(scale, 1.0) * 0 + 1e-12 * max(scale, 1.0)
        for i in np.nonzero(~ok)[0][:3]:
            bad.append(f"{name} roi {i}: got {g[i]!r} want {w[i]!r}")
    return bad
