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


def write_tiled_tiff(path: str, arr: np.ndarray, tile: int = 1024, strips: bool = False, rows_per_strip: int = 64) -> None:
    """Minimal classic-TIFF writer for tests: one grayscale page of uint8 / uint16 / uint32 samples, stored as deflate-compressed
    TILES (default; what the reference's NyxusGrayscaleTiffTileLoader reads) or STRIPS (NyxusGrayscaleTiffStripLoader)."""
    import struct
    import zlib
    a = np.ascontiguousarray(arr)
    h, w = a.shape
    bits = a.dtype.itemsize * 8
    chunks = []
    if strips:
        for y in range(0, h, rows_per_strip):
            chunks.append(zlib.compress(a[y:y + rows_per_strip].tobytes(), 6))
    else:
        for y in range(0, h, tile):
            for x in range(0, w, tile):
                t = np.zeros((tile, tile), a.dtype)
                blk = a[y:y + tile, x:x + tile]
                t[:blk.shape[0], :blk.shape[1]] = blk
                chunks.append(zlib.compress(t.tobytes(), 6))
    n = len(chunks)
    tags = [(256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, bits), (259, 3, 1, 8), (262, 3, 1, 1), (277, 3, 1, 1), (284, 3, 1, 1), (339, 3, 1, 1)]
    if strips:
        tags += [(278, 4, 1, rows_per_strip), (273, 4, n, None), (279, 4, n, None)]
    else:
        tags += [(322, 4, 1, tile), (323, 4, 1, tile), (324, 4, n, None), (325, 4, n, None)]
    tags.sort()
    ifd_off = 8
    ifd_size = 2 + 12 * len(tags) + 4
    arr_off = ifd_off + ifd_size                  # the two LONG arrays (offsets, byte counts), then the data
    data_off = arr_off + 8 * n
    offs, pos = [], data_off
    for c in chunks:
        offs.append(pos)
        pos += len(c)
    with open(path, "wb") as fh:
        fh.write(struct.pack("<2sHI", b"II", 42, ifd_off))
        fh.write(struct.pack("<H", len(tags)))
        for tag, typ, cnt, val in tags:
            if val is None:                       # offsets first, byte counts second
                first = tag in (273, 324)
                v = (arr_off if first else arr_off + 4 * n) if n > 1 else (offs[0] if first else len(chunks[0]))
                fh.write(struct.pack("<HHII", tag, typ, cnt, v))
            elif typ == 3:
                fh.write(struct.pack("<HHIHH", tag, typ, cnt, val, 0))
            else:
                fh.write(struct.pack("<HHII", tag, typ, cnt, val))
        fh.write(struct.pack("<I", 0))
        fh.write(struct.pack("<%dI" % n, *offs))
        fh.write(struct.pack("<%dI" % n, *[len(c) for c in chunks]))
        for c in chunks:
            fh.write(c)


def slide4096(seed: int = 0, size: int = 4096, pitch: int = 256):
    """A synthetic slide for the tiled-TIFF path: blocky 12-bit intensities (8 x 8 blocks, so that the deflate tiles stay small)
    with a little per-pixel structure, and a 16 x 16 grid of disks of radius 20..110 -- many cross the 1024-pixel TIFF tile
    borders; label values are sparse (multiples of 1000)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size]
    blk = rng.integers(1, 4000, (size // 8, size // 8)).astype(np.uint32)
    inten = (np.kron(blk, np.ones((8, 8), np.uint32)) + ((xx * 3 + yy * 5) % 17)).astype(np.uint16)
    lab = np.zeros((size, size), np.uint32)
    k = 0
    for gy in range(size // pitch):
        for gx in range(size // pitch):
            k += 1
            cy = gy * pitch + pitch // 2 + int(rng.integers(-10, 11))
            cx = gx * pitch + pitch // 2 + int(rng.integers(-10, 11))
            r = int(rng.integers(20, 111))
            y0, y1, x0, x1 = max(cy - r, 0), min(cy + r + 1, size), max(cx - r, 0), min(cx + r + 1, size)
            m = (yy[y0:y1, x0:x1] - cy) ** 2 + (xx[y0:y1, x0:x1] - cx) ** 2 <= r * r
            lab[y0:y1, x0:x1][m] = 1000 * k
    return inten, lab
