"""Deterministic synthetic inputs shared by tests, bench.py and smoke().

Tile recipe = SURVEY.md section 8(d) / BASELINE.md section 3: 1024x1024 uint32 tile,
intensities uniform in [1, 4095], label image = 14x14 grid of disks (pitch 73 px,
radius 30 px -> 196 ROIs of 2821 px, bbox 61x61); an "irregular" variant with
per-ROI radius in [8, 36) and 10 % concave ROIs.
"""
from __future__ import annotations

import numpy as np

from nyxus_amd import _abi


def disk_label_tile(size: int = 1024, pitch: int = 73, radius: int = 30, irregular: bool = False, seed: int = 0):
    rng = np.random.default_rng(seed)
    lab = np.zeros((size, size), np.uint32)
    n_side = size // pitch
    yy, xx = np.mgrid[-pitch // 2:pitch // 2 + 1, -pitch // 2:pitch // 2 + 1]
    k = 0
    for gy in range(n_side):
        for gx in range(n_side):
            k += 1
            cy, cx = gy * pitch + pitch // 2, gx * pitch + pitch // 2
            r = int(rng.integers(8, 36)) if irregular else radius
            m = (xx * xx + yy * yy) <= r * r
            if irregular and rng.random() < 0.1:
                m &= ~(((xx - r // 2) ** 2 + (yy - r // 2) ** 2) <= (r // 2) ** 2)  # bite -> concave
            y0, x0 = cy - pitch // 2, cx - pitch // 2
            sub = lab[max(y0, 0):y0 + m.shape[0], max(x0, 0):x0 + m.shape[1]]
            mm = m[max(-y0, 0):max(-y0, 0) + sub.shape[0], max(-x0, 0):max(-x0, 0) + sub.shape[1]]
            sub[mm] = k
    return lab


def intensity_tile(k: int, size: int = 1024, lo: int = 1, hi: int = 4096):
    return np.random.default_rng(1234 + k).integers(lo, hi, (size, size), dtype=np.uint32)


def rois_from_tile(inten: np.ndarray, lab: np.ndarray):
    """Host restatement of phases 1-2 of the in-memory workflow
    (/root/reference/src/nyx/phase1.cpp:373-409, phase2_2d.cpp:637-684): per label
    -> pixel cloud in COLUMN-major scan order, min/max, bounding box.
    Vectorised with one stable argsort over the labels."""
    # column-major scan order: x outer, y inner
    labT = lab.T.ravel()
    intT = inten.T.ravel()
    h, w = lab.shape
    idx = np.nonzero(labT)[0]
    order = np.argsort(labT[idx], kind="stable")
    idx = idx[order]
    labs = labT[idx]
    xs = (idx // h).astype(np.int64)
    ys = (idx % h).astype(np.int64)
    vals = intT[idx]
    uniq, start = np.unique(labs, return_index=True)
    bounds = np.append(start, len(labs))
    rois = []
    for i, l in enumerate(uniq):
        a, b = bounds[i], bounds[i + 1]
        rois.append(dict(label=int(l), x=xs[a:b], y=ys[a:b], inten=vals[a:b]))
    return rois


def tile_batch(k: int = 0, irregular: bool = False, size: int = 1024, lo: int = 1, hi: int = 4096) -> _abi.HostBatch:
    lab = disk_label_tile(size=size, irregular=irregular, seed=k)
    it = intensity_tile(k, size=size, lo=lo, hi=hi)
    return _abi.batch_from_rois(rois_from_tile(it, lab))


def random_rois(n_roi: int, seed: int = 0, rmax: int = 25, value_modes=(4096, 256, 65536, 8, 2 ** 32 - 1), slide=False):
    """Small irregular ROIs covering the edge cases the reference tests exercise:
    constant ROIs, all-zero ROIs, zero-valued pixels inside the ROI, holes in the bbox,
    single-pixel ROIs, full-range uint32 values."""
    rng = np.random.default_rng(seed)
    rois = []
    for k in range(n_roi):
        r = int(rng.integers(0, rmax))
        yy, xx = np.mgrid[-r:r + 1, -r:r + 1]
        m = ((xx * xx + yy * yy) <= r * r) & (rng.random(xx.shape) > 0.1)
        if m.sum() == 0:
            m[r, r] = True
        y, x = np.nonzero(m)
        hi = value_modes[k % len(value_modes)]
        v = rng.integers(0 if k % 5 == 0 else 1, hi, len(x)).astype(np.uint32)
        if k % 11 == 3:
            v[:] = 7
        if k % 13 == 5:
            v[:] = 0
        o = np.lexsort((y, x))
        d = dict(x=x[o], y=y[o], inten=v[o], label=k + 1)
        if slide:
            d["slide_min"] = 0.0
            d["slide_max"] = float(2 ** 16)
        rois.append(d)
    return rois
