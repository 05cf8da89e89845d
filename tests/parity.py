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


def compare_tables(got: np.ndarray, want: np.ndarray, names, rel=REL_TOL, exact=EXACT_COLUMNS, atol=None):
    """Returns a list of human-readable mismatches (empty = parity).
    atol: optional {column name: per-row absolute tolerance} for columns that are zero up to cancellation noise by
    construction (first-order central moments), where neither a relative bound nor the column scale means anything."""
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
        if base in ("GLCM_INFOMEAS2", "GLCM_INFOMEAS2_AVE"):
            # sqrt(|1 - exp(-2 (HXY2 - HXY))|): when the matrix is (numerically) a product of its marginals the argument is
            # +-1 ulp of cancellation noise and the value is 0 or sqrt(2^-52 .. 2^-50) = 1.5e-8 .. 3e-8 on either side
            # (the reference itself lands on both); on the feature's [0, 1] scale that is an absolute 1e-7.
            with np.errstate(invalid="ignore"):
                ok |= np.abs(g - w) <= 1e-7
        if atol is not None and name in atol:
            with np.errstate(invalid="ignore"):
                ok |= np.abs(g - w) <= np.asarray(atol[name])
        for i in np.nonzero(~ok)[0][:3]:
            bad.append(f"{name} roi {i}: got {g[i]!r} want {w[i]!r}")
    return bad


def moment_atol(b):
    """Per-row absolute tolerances for the first-order central moments of a HostBatch: they vanish identically
    (sum I (x - m10/m00)); what is left is rounding noise of magnitude eps * sum I |x - cx|.  Bound: 1e-9 * m00 * side."""
    off = np.asarray(b.px_offset).astype(np.int64)
    m00_s = np.diff(off).astype(float)
    m00_i = np.array([np.asarray(b.inten[off[r]:off[r + 1]], dtype=float).sum() for r in range(len(m00_s))])
    side = np.maximum(b.bbox_w, b.bbox_h).astype(float)
    tol = {"CENTRAL_MOMENT_01": 1e-9 * m00_s * side, "CENTRAL_MOMENT_10": 1e-9 * m00_s * side,
           "IMOM_CM_01": 1e-9 * m00_i * side, "IMOM_CM_10": 1e-9 * m00_i * side}
    # Higher central moments: sums of m00 terms of size up to (side / 2)^(p + q) that cancel for symmetric shapes (odd orders
    # of an ellipse vanish identically); what survives is rounding noise ~ eps * sum |terms|.  Floor: 1e-13 of that sum's
    # bound -- eight orders below the 1e-5 relative bound that applies whenever the moment is not a cancellation.
    for p in range(4):
        for q in range(4):
            if p + q < 2:
                continue
            scale = 1e-13 * (side / 2.0) ** (p + q)
            tol["CENTRAL_MOMENT_%d%d" % (p, q)] = scale * m00_s
            tol["IMOM_CM_%d%d" % (p, q)] = scale * m00_i
    return tol
