"""Loaders for the committed golden fixtures (tests/golden/)."""
from __future__ import annotations

import json
import os

import numpy as np

from nyxus_amd import _abi

_G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def reference_tests():
    with open(os.path.join(_G, "reference_tests.json")) as fh:
        return json.load(fh)


def roi_from_triplets(trip, mask_trip=None):
    """{x,y,intensity} triplets (+ optional same-shaped mask triplets) -> ROI dict, the
    way /root/reference/tests/test_main_nyxus.h:26-88 feeds them to an LR."""
    t = np.asarray(trip, np.int64)
    keep = np.ones(len(t), bool) if mask_trip is None else (np.asarray(mask_trip, np.int64)[:, 2] != 0)
    t = t[keep]
    return dict(x=t[:, 0], y=t[:, 1], inten=t[:, 2].astype(np.uint32))


def dsb_roi(d):
    """DSB2018 fixture: every pixel of the w*h frame enters the ROI (zeros included),
    /root/reference/tests/test_main_nyxus.h:90-124."""
    w, h = d["w"], d["h"]
    v = np.asarray(d["pixels"], np.uint32)
    i = np.arange(w * h)
    return dict(x=i % w, y=i // w, inten=v)


def ibsi_phantom_batch(ref):
    rois = [roi_from_triplets(ref["pixels"][f"ibsi_phantom_z{z}_intensity"], ref["pixels"][f"ibsi_phantom_z{z}_mask"])
            for z in range(1, 5)]
    return _abi.batch_from_rois(rois)


# ---- 2-D moment fixtures: recipes of /root/reference/tests/test_2d_moments_common.h:37-72 (48 x 40 rectangle with a
# position-dependent intensity) and :92-118 (thin right wedge 5*y <= x of unit intensity) --------------------------------
SMOM_NAMES = (["SPAT_MOMENT_%d%d" % pq for pq in [(0, 0), (0, 1), (0, 2), (0, 3), (1, 0), (1, 1), (1, 2), (1, 3), (2, 0), (2, 1), (2, 2), (2, 3), (3, 0)]]
              + ["CENTRAL_MOMENT_%d%d" % (p, q) for p in range(4) for q in range(4)]
              + ["NORM_SPAT_MOMENT_%d%d" % (p, q) for p in range(4) for q in range(4)]
              + ["NORM_CENTRAL_MOMENT_%s" % k for k in ("02", "03", "11", "12", "20", "21", "30")]
              + ["HU_M%d" % k for k in range(1, 8)]
              + ["WEIGHTED_SPAT_MOMENT_%s" % k for k in ("00", "01", "02", "03", "10", "11", "12", "20", "21", "30")]
              + ["WEIGHTED_CENTRAL_MOMENT_%s" % k for k in ("02", "03", "11", "12", "20", "21", "30")]
              + ["WT_NORM_CTR_MOM_%s" % k for k in ("02", "03", "11", "12", "20", "21", "30")]
              + ["WEIGHTED_HU_M%d" % k for k in range(1, 8)])
IMOM_NAMES = (["IMOM_RM_%d%d" % pq for pq in [(0, 0), (0, 1), (0, 2), (0, 3), (1, 0), (1, 1), (1, 2), (1, 3), (2, 0), (2, 1), (2, 2), (2, 3), (3, 0)]]
              + ["IMOM_CM_%d%d" % (p, q) for p in range(4) for q in range(4)]
              + ["IMOM_NRM_%d%d" % (p, q) for p in range(4) for q in range(4)]
              + ["IMOM_NCM_%s" % k for k in ("02", "03", "11", "12", "20", "21", "30")]
              + ["IMOM_HU%d" % k for k in range(1, 8)]
              + ["IMOM_WRM_%s" % k for k in ("00", "01", "02", "03", "10", "11", "12", "20", "21", "30")]
              + ["IMOM_WCM_%s" % k for k in ("02", "03", "11", "12", "20", "21", "30")]
              + ["IMOM_WNCM_%s" % k for k in ("02", "03", "11", "12", "20", "21", "30")]
              + ["IMOM_WHU%d" % k for k in range(1, 8)])


def geomoment_rectangle_roi():
    yy, xx = np.mgrid[0:40, 0:48]
    inten = (10.0 + 3.0 * xx + 5.0 * yy + ((xx * yy) % 7)).astype(np.uint32)
    return dict(x=xx.ravel(), y=yy.ravel(), inten=inten.ravel())      # row-major push order of the fixture


def geomoment_wedge_roi():
    yy, xx = np.mgrid[0:8, 0:40]
    keep = (5 * yy <= xx)
    return dict(x=xx[keep], y=yy[keep], inten=np.ones(int(keep.sum()), np.uint32))
