"""Host-side ROI assembly for the in-memory API: what phases 1-2 of the reference do before the hot path.

  phase 1  gatherRoisMetricsInMemory   /root/reference/src/nyx/phase1.cpp:373-409
           feed_pixel_2_metrics        src/nyx/pixel_feed.cpp:19-43   (per label: area, min, max, AABB)
  phase 2  scanTrivialRoisInMemory     src/nyx/phase2_2d.cpp:637-684  (pixel clouds, COLUMN-major scan)
           allocateTrivialRoisBuffers  src/nyx/phase2_2d.cpp:427-465  (dense plane -- rebuilt in LDS here)

Vectorised NumPy (one stable argsort over the labels) producing the SoA batch of the C ABI; rows come
out in ascending label order, the row order of save_features_2_buffer (output_2_buffer.cpp:305-306).
The device version of this step is nyxhip_featurize_tile.
"""
from __future__ import annotations

import numpy as np

from nyxus_amd import _abi


def assemble(inten: np.ndarray, label: np.ndarray, slide_min=None, slide_max=None) -> _abi.HostBatch:
    if inten.shape != label.shape or inten.ndim != 2:
        raise ValueError("intensity and label tiles must be 2-D arrays of the same shape")
    h, w = label.shape
    labT = np.ascontiguousarray(label.T).ravel()          # column-major scan order: x outer, y inner
    intT = np.ascontiguousarray(inten.T).ravel()
    idx = np.flatnonzero(labT)
    if idx.size == 0:
        return None
    order = np.argsort(labT[idx], kind="stable")
    idx = idx[order]
    labs = labT[idx]
    xs = idx // h
    ys = idx % h
    vals = intT[idx].astype(np.uint32)
    uniq, start = np.unique(labs, return_index=True)
    bounds = np.append(start, len(labs)).astype(np.int64)
    xmin = np.minimum.reduceat(xs, start)
    xmax = np.maximum.reduceat(xs, start)
    ymin = np.minimum.reduceat(ys, start)
    ymax = np.maximum.reduceat(ys, start)
    vmin = np.minimum.reduceat(vals, start)
    vmax = np.maximum.reduceat(vals, start)
    counts = np.diff(bounds)
    if np.any(xmax - xmin > 65535) or np.any(ymax - ymin > 65535):
        raise ValueError("an ROI's bounding box is wider or taller than 65536 pixels: coordinates inside a box are 16-bit (NYXHIP_ERR_ROI_TOO_LARGE)")
    rel_x = (xs - np.repeat(xmin, counts)).astype(np.uint16)
    rel_y = (ys - np.repeat(ymin, counts)).astype(np.uint16)
    n = len(uniq)
    smin = None if slide_min is None else np.full(n, slide_min, np.float64)
    smax = None if slide_max is None else np.full(n, slide_max, np.float64)
    return _abi.HostBatch(uniq.astype(np.uint32), bounds.astype(np.uint64), rel_x, rel_y, vals,
                          (xmax - xmin + 1).astype(np.uint32), (ymax - ymin + 1).astype(np.uint32),
                          vmin.astype(np.uint32), vmax.astype(np.uint32), smin, smax)
