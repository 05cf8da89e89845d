"""Comparison policy shared by the parity tests (north_star):
bit-exact for label/count features, <= 1e-5 relative otherwise.  The only absolute allowance a column gets by default is
1e-12 of the column's own scale (its largest |value| in the expected table); columns that are sums with cancellation
(central / normalised central / Hu moments) get per-row floors DERIVED from the size of the cancelling terms (moment_atol),
never a floor tied to 1.0 -- so a column whose values are 1e-30 is still checked at 1e-5 of ITS values."""
from __future__ import annotations

import numpy as np

REL_TOL = 1e-5
# integer-exact columns: sums / order statistics of integers and integer-valued ratios
EXACT_COLUMNS = {"MIN", "MAX", "RANGE", "MODE", "MEDIAN", "INTEGRATED_INTENSITY", "ENERGY", "MEAN",
                 "ROOT_MEAN_SQUARED", "P01", "P10", "P25", "P75", "P90", "P99", "INTERQUARTILE_RANGE", "QCOD",
                 "UNIFORMITY_PIU", "COVERED_IMAGE_INTENSITY_RANGE", "ROBUST_MEAN"}


# rows that passed a column only through an absolute floor LARGER than the expected value itself -- a vacuous check; compare_tables
# counts them per column here (reset by the caller) and reports a column as a mismatch when more than WAIVE_LIMIT of its rows pass
# that way: a floor that swallows the column is a defect of the floor, not parity
WAIVED = {}
WAIVE_LIMIT = 0.34
# (the first-order central moments vanish identically -- sum I (x - m10 / m00) -- so their expected values ARE rounding noise and the
#  floor, eps-sized against the cancelling terms, is the whole check: exempt from the limit, still counted)
IDENTICALLY_ZERO = {"CENTRAL_MOMENT_01", "CENTRAL_MOMENT_10", "IMOM_CM_01", "IMOM_CM_10"}


def compare_tables(got: np.ndarray, want: np.ndarray, names, rel=REL_TOL, exact=EXACT_COLUMNS, atol=None, batch=None):
    """Returns a list of human-readable mismatches (empty = parity).
    atol: optional {column name: per-row absolute tolerance} for columns that are zero up to cancellation noise by
    construction (central moments and what is derived from them), where neither a relative bound nor the column scale means
    anything.  batch: the HostBatch the table was computed from -- the floors of the moment columns are then derived here
    (moment_atol) when the table holds any."""
    assert got.shape == want.shape, (got.shape, want.shape)
    if atol is None and batch is not None and any(n.startswith(("CENTRAL_MOMENT_", "IMOM_CM_")) for n in names):
        atol = moment_atol(batch, want, names)
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
                ok |= np.abs(g - w) <= rel * np.abs(w) + 1e-12 * scale
        if base in ("GLCM_INFOMEAS2", "GLCM_INFOMEAS2_AVE"):
            # sqrt(|1 - exp(-2 (HXY2 - HXY))|): when the matrix is (numerically) a product of its marginals the argument is
            # +-1 ulp of cancellation noise and the value is 0 or sqrt(2^-52 .. 2^-50) = 1.5e-8 .. 3e-8 on either side
            # (the reference itself lands on both); on the feature's [0, 1] scale that is an absolute 1e-7.
            with np.errstate(invalid="ignore"):
                ok |= np.abs(g - w) <= 1e-7
        if atol is not None and name in atol:
            fl = np.broadcast_to(np.asarray(atol[name], dtype=float), g.shape)
            with np.errstate(invalid="ignore"):
                by_floor = ~ok & (np.abs(g - w) <= fl)
                vacuous = by_floor & ~(fl <= np.abs(w))                  # floor above |want| (or infinite): nothing was checked
            ok |= by_floor
            if vacuous.any():
                WAIVED[name] = WAIVED.get(name, 0) + int(vacuous.sum())
                if name not in IDENTICALLY_ZERO and len(g) >= 32 and vacuous.sum() > WAIVE_LIMIT * len(g):   # (small samples of symmetric shapes: odd moments vanish)
                    bad.append(f"{name}: {int(vacuous.sum())} of {len(g)} rows pass only through a floor larger than the expected value")
        for i in np.nonzero(~ok)[0][:3]:
            bad.append(f"{name} roi {i}: got {g[i]!r} want {w[i]!r}")
    return bad


def tolerance_of(name, w, rel=REL_TOL, exact=EXACT_COLUMNS, atol=None):
    """The per-row tolerance compare_tables grants column `name` with expected values w (0 for exact columns): for margin reports."""
    base = name
    for suf in ("_0", "_45", "_90", "_135"):
        if base.endswith(suf):
            base = base[: -len(suf)]
    if base in exact:
        return np.zeros(len(w))
    scale = np.nanmax(np.abs(np.where(np.isfinite(w), w, 0.0))) if len(w) else 0.0
    tol = rel * np.abs(np.where(np.isfinite(w), w, 0.0)) + 1e-12 * scale
    if base in ("GLCM_INFOMEAS2", "GLCM_INFOMEAS2_AVE"):
        tol = np.maximum(tol, 1e-7)
    if atol is not None and name in atol:
        tol = np.maximum(tol, np.asarray(atol[name]))
    return tol


_NC = ("02", "03", "11", "12", "20", "21", "30")


def _hu7(e):
    """calcHu_imp (/root/reference/src/nyx/features/2d_geomoments_basic.cpp:231-253) on arrays; e: dict pq -> eta_pq."""
    _02, _03, _11, _12, _20, _21, _30 = (e[k] for k in _NC)
    a, b = _30 + _12, _21 + _03
    return [_20 + _02,
            (_20 - _02) ** 2 + 4 * _11 ** 2,
            (_30 - 3 * _12) ** 2 + (3 * _21 - _03) ** 2,
            a ** 2 + b ** 2,
            (_30 - 3 * _12) * a * (a ** 2 - 3 * b ** 2) + (3 * _21 - _03) * b * (3 * a ** 2 - b ** 2),
            (_20 - _02) * (a ** 2 - b ** 2) + 4 * _11 * a * b,
            (3 * _21 - _03) * a * (a ** 2 - 3 * b ** 2) - (_30 - 3 * _12) * b * (3 * a ** 2 - b ** 2)]


def _hu_floor(eta, d_eta):
    """First-order propagation of the allowance d_eta on the seven normalised central moments through the Hu polynomials
    (sum over inputs of |H(eta + d e_k) - H(eta)|), plus 1e-13 of the magnitude the polynomial's terms reach when every
    eta is replaced by |eta| (rounding of the evaluation itself)."""
    h0 = _hu7(eta)
    out = [np.zeros_like(h0[0]) for _ in range(7)]
    for k in _NC:
        e2 = dict(eta)
        e2[k] = eta[k] + d_eta[k]
        for i, (hp, hb) in enumerate(zip(_hu7(e2), h0)):
            out[i] = out[i] + np.abs(hp - hb)
    # magnitude of the terms: evaluate with |eta| and every subtraction turned into an addition
    A = {k: np.abs(v) for k, v in eta.items()}
    a, b = A["30"] + A["12"], A["21"] + A["03"]
    mag = [A["20"] + A["02"], (A["20"] + A["02"]) ** 2 + 4 * A["11"] ** 2, (A["30"] + 3 * A["12"]) ** 2 + (3 * A["21"] + A["03"]) ** 2,
           a ** 2 + b ** 2, (A["30"] + 3 * A["12"]) * a * (a ** 2 + 3 * b ** 2) + (3 * A["21"] + A["03"]) * b * (3 * a ** 2 + b ** 2),
           (A["20"] + A["02"]) * (a ** 2 + b ** 2) + 4 * A["11"] * a * b,
           (3 * A["21"] + A["03"]) * a * (a ** 2 + 3 * b ** 2) + (A["30"] + 3 * A["12"]) * b * (3 * a ** 2 + b ** 2)]
    return [o + 1e-13 * m for o, m in zip(out, mag)]


def moment_atol(b, want=None, names=None):
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
    if want is None:
        return tol
    # ---- columns derived from the central moments: their floors follow from the floors above -------------------------------
    # (names / want: the expected table, from which the normalising masses and the eta_pq are read)
    col = {n: want[:, j] for j, n in enumerate(names)}
    IN_REL = 1e-9                                     # allowance on an eta beyond its own cancellation floor
    fam = [("CENTRAL_MOMENT_", "NORM_CENTRAL_MOMENT_", "HU_M", "SPAT_MOMENT_00", "WEIGHTED_SPAT_MOMENT_", "WEIGHTED_CENTRAL_MOMENT_", "WT_NORM_CTR_MOM_", "WEIGHTED_HU_M", m00_s),
           ("IMOM_CM_", "IMOM_NCM_", "IMOM_HU", "IMOM_RM_00", "IMOM_WRM_", "IMOM_WCM_", "IMOM_WNCM_", "IMOM_WHU", m00_i)]
    for cm, ncm, hu, m00n, wrm, wcm, wncm, whu, mass in fam:
        if m00n not in col:
            continue
        m00 = np.abs(col[m00n])
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            eta, d_eta = {}, {}
            for k in _NC:                                                   # normCentralMom :212-217: CM_pq / m00^((p+q)/2 + 1)
                t = (int(k[0]) + int(k[1])) / 2.0 + 1.0
                tol[ncm + k] = tol[cm + k] / m00 ** t
                eta[k] = col[ncm + k]
                d_eta[k] = IN_REL * np.abs(eta[k]) + tol[ncm + k]
            for i, f in enumerate(_hu_floor(eta, d_eta)):
                tol[hu + str(i + 1)] = f
            # weighted set: weights I log(d + 0.001), |log| <= 7 for boxes below 1000 px (contour pixels: log(0.001) = -6.9); the
            # weighted origin (w10 / w00, w01 / w00) may lie far outside the box when the weighted mass nearly cancels
            w00 = np.abs(col[wrm + "00"])
            ox, oy = np.abs(col[wrm + "10"]) / w00, np.abs(col[wrm + "01"]) / w00
            reach = side + ox + oy
            weta, d_weta = {}, {}
            for k in _NC:
                pq = int(k[0]) + int(k[1])
                tol[wcm + k] = 1e-13 * 7.0 * mass * reach ** pq
                tol[wncm + k] = tol[wcm + k] / w00 ** (pq / 2.0 + 1.0)
                weta[k] = col[wncm + k]
                d_weta[k] = IN_REL * np.abs(weta[k]) + tol[wncm + k]
            for i, f in enumerate(_hu_floor(weta, d_weta)):
                tol[whu + str(i + 1)] = f
    return {k: np.where(np.isfinite(v), v, np.inf) for k, v in tol.items()}
