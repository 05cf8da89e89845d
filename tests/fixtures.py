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
