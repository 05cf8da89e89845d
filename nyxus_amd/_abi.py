"""ctypes mirror of ``include/nyxhip.h`` (struct layouts, constants) and small
helpers that pack NumPy arrays / device pointers into a ``nyxhip_batch``.

This module only describes the C ABI; it loads nothing.  ``_lib.py`` loads the
HIP library and fails loudly when it is missing (there is no CPU fallback).
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

NYXHIP_OK = 0
ERR_NAMES = {0: "OK", 1: "INVALID_ARG", 2: "NO_DEVICE", 3: "HIP", 4: "UNSUPPORTED", 5: "ROI_TOO_LARGE"}

FAM_INTENSITY = 1 << 0
FAM_GLCM = 1 << 1
FAM_GLRLM = 1 << 2
FAM_GLSZM = 1 << 3
FAM_NGTDM = 1 << 4
FAM_GABOR = 1 << 5
FAM_ZERNIKE = 1 << 6
FAM_GLDZM = 1 << 7
FAM_GLDM = 1 << 8
FAM_NGLDM = 1 << 9
FAM_SMOMS = 1 << 10
FAM_IMOMS = 1 << 11
FAM_NORTH_STAR = 0x7F
FAM_ALL = 0xFFF

MEM_HOST = 0
MEM_DEVICE = 1
MEM_HOST_OWN_MAPPING = 2   # nyxhip_tiles only: host arrays the caller declares mappings of their own (registered for DMA in place)

U8, U16, U32 = 1, 2, 4                       # tile element types (bytes per element)
SLIDE_MONTAGE, SLIDE_PER_TILE, SLIDE_GIVEN = 0, 1, 2

MAX_GLCM_ANGLES = 4
MAX_GABOR_FILTERS = 16


class Settings(C.Structure):
    """``nyxhip_settings`` (include/nyxhip.h)."""

    _fields_ = [
        ("soft_nan", C.c_double),
        ("tiny", C.c_double),
        ("grey_depth", C.c_int32),
        ("ibsi", C.c_int32),
        ("glcm_grey_depth", C.c_int32),
        ("glcm_offset", C.c_int32),
        ("glcm_n_angles", C.c_int32),
        ("glcm_angles", C.c_int32 * MAX_GLCM_ANGLES),
        ("glcm_symmetric", C.c_int32),
        ("gabor_gamma", C.c_double),
        ("gabor_sig2lam", C.c_double),
        ("gabor_f0lp", C.c_double),
        ("gabor_graythr", C.c_double),
        ("gabor_kersize", C.c_int32),
        ("gabor_n_filters", C.c_int32),
        ("gabor_f0", C.c_double * MAX_GABOR_FILTERS),
        ("gabor_theta", C.c_double * MAX_GABOR_FILTERS),
    ]


class Batch(C.Structure):
    """``nyxhip_batch`` (include/nyxhip.h)."""

    _fields_ = [
        ("n_roi", C.c_uint64),
        ("roi_label", C.c_void_p),
        ("px_offset", C.c_void_p),
        ("x", C.c_void_p),
        ("y", C.c_void_p),
        ("inten", C.c_void_p),
        ("bbox_w", C.c_void_p),
        ("bbox_h", C.c_void_p),
        ("min_inten", C.c_void_p),
        ("max_inten", C.c_void_p),
        ("slide_min", C.c_void_p),
        ("slide_max", C.c_void_p),
        ("memory", C.c_int32),
        ("max_px", C.c_uint32),
        ("max_bbox_area", C.c_uint32),
        ("max_inten_range", C.c_uint32),
        ("max_bbox_side", C.c_uint32),
    ]


class Tiles(C.Structure):
    """``nyxhip_tiles`` (include/nyxhip.h)."""

    _fields_ = [
        ("inten", C.c_void_p),
        ("label", C.c_void_p),
        ("inten_dtype", C.c_int32),
        ("label_dtype", C.c_int32),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("n_tiles", C.c_uint32),
        ("memory", C.c_int32),
        ("slide_mode", C.c_int32),
        ("slide_min", C.c_void_p),
        ("slide_max", C.c_void_p),
        ("max_device_bytes", C.c_uint64),
    ]


def default_settings(coarse_gray_depth: int = 64, ibsi: bool = False) -> Settings:
    """Reference defaults: Environment::compile_feature_settings
    (/root/reference/src/nyx/env_features.cpp:713-736), glcm.cpp:8-9, gabor.cpp:14-25."""
    s = Settings()
    s.soft_nan = 0.0
    s.tiny = 1e-10
    s.grey_depth = int(coarse_gray_depth)
    s.ibsi = 1 if ibsi else 0
    s.glcm_grey_depth = int(coarse_gray_depth)
    s.glcm_offset = 1
    s.glcm_n_angles = 4
    for i, a in enumerate((0, 45, 90, 135)):
        s.glcm_angles[i] = a
    s.glcm_symmetric = 0
    s.gabor_gamma = 0.1
    s.gabor_sig2lam = 0.8
    s.gabor_f0lp = 0.1
    s.gabor_graythr = 0.025
    s.gabor_kersize = 16
    s.gabor_n_filters = 4
    # gabor.cpp:19-25 -- the pairs are declared {theta-like, f0-like} but consumed as
    # (first=f0, second=theta) at gabor.cpp:107-110; keep the consumed meaning.
    for i, (f0, th) in enumerate(((0.0, 4.0), (math.pi / 4, 16.0), (math.pi / 2, 32.0), (math.pi / 4 * 3.0, 64.0))):
        s.gabor_f0[i] = f0
        s.gabor_theta[i] = th
    return s


@dataclass
class HostBatch:
    """A host-memory ROI batch: keeps the NumPy arrays alive next to the C struct."""

    roi_label: np.ndarray
    px_offset: np.ndarray
    x: np.ndarray
    y: np.ndarray
    inten: np.ndarray
    bbox_w: np.ndarray
    bbox_h: np.ndarray
    min_inten: np.ndarray
    max_inten: np.ndarray
    slide_min: Optional[np.ndarray] = None
    slide_max: Optional[np.ndarray] = None

    def __post_init__(self):
        self.roi_label = np.ascontiguousarray(self.roi_label, np.uint32)
        self.px_offset = np.ascontiguousarray(self.px_offset, np.uint64)
        self.x = np.ascontiguousarray(self.x, np.uint16)
        self.y = np.ascontiguousarray(self.y, np.uint16)
        self.inten = np.ascontiguousarray(self.inten, np.uint32)
        self.bbox_w = np.ascontiguousarray(self.bbox_w, np.uint32)
        self.bbox_h = np.ascontiguousarray(self.bbox_h, np.uint32)
        self.min_inten = np.ascontiguousarray(self.min_inten, np.uint32)
        self.max_inten = np.ascontiguousarray(self.max_inten, np.uint32)
        if self.slide_min is not None:
            self.slide_min = np.ascontiguousarray(self.slide_min, np.float64)
            self.slide_max = np.ascontiguousarray(self.slide_max, np.float64)
        n = len(self.roi_label)
        if len(self.px_offset) != n + 1:
            raise ValueError("px_offset must have n_roi+1 entries")
        if int(self.px_offset[-1]) != len(self.inten) or len(self.x) != len(self.inten) or len(self.y) != len(self.inten):
            raise ValueError("x / y / inten length must equal px_offset[-1]")

    @property
    def n_roi(self) -> int:
        return len(self.roi_label)

    @property
    def n_px(self) -> int:
        return len(self.inten)

    def c_struct(self) -> Batch:
        b = Batch()
        b.n_roi = self.n_roi
        for name in ("roi_label", "px_offset", "x", "y", "inten", "bbox_w", "bbox_h", "min_inten", "max_inten"):
            setattr(b, name, getattr(self, name).ctypes.data)
        b.slide_min = self.slide_min.ctypes.data if self.slide_min is not None else None
        b.slide_max = self.slide_max.ctypes.data if self.slide_max is not None else None
        b.memory = MEM_HOST
        b.max_px = int(np.diff(self.px_offset.astype(np.int64)).max()) if self.n_roi else 0
        b.max_bbox_area = int((self.bbox_w.astype(np.int64) * self.bbox_h.astype(np.int64)).max()) if self.n_roi else 0
        b.max_inten_range = int((self.max_inten.astype(np.int64) - self.min_inten.astype(np.int64)).max()) if self.n_roi else 0
        b.max_bbox_side = int(max(self.bbox_w.max(), self.bbox_h.max())) if self.n_roi else 0
        return b


def batch_from_rois(rois: Sequence[dict]) -> HostBatch:
    """Builds a HostBatch from a list of dicts with keys x, y (absolute or relative
    coordinates), inten and optionally label / slide_min / slide_max.  Coordinates
    are re-based to the ROI's bounding box; min/max come from the pixels
    (what phase 1 of the reference records in LR::aux_min/aux_max,
    /root/reference/src/nyx/pixel_feed.cpp:19-43)."""
    labels, offs, xs, ys, it, bw, bh, mn, mx = [], [0], [], [], [], [], [], [], []
    smin, smax = [], []
    for k, r in enumerate(rois):
        x = np.asarray(r["x"], np.int64)
        y = np.asarray(r["y"], np.int64)
        v = np.asarray(r["inten"], np.uint32)
        if len(v) == 0:
            raise ValueError("empty ROI")
        x0, y0 = x.min(), y.min()
        xs.append((x - x0).astype(np.uint16))
        ys.append((y - y0).astype(np.uint16))
        it.append(v)
        bw.append(int(x.max() - x0 + 1))
        bh.append(int(y.max() - y0 + 1))
        mn.append(int(r.get("min", v.min())))
        mx.append(int(r.get("max", v.max())))
        labels.append(int(r.get("label", k + 1)))
        offs.append(offs[-1] + len(v))
        if "slide_min" in r:
            smin.append(float(r["slide_min"]))
            smax.append(float(r["slide_max"]))
    has_slide = len(smin) == len(rois) and len(rois) > 0
    return HostBatch(
        np.array(labels), np.array(offs), np.concatenate(xs), np.concatenate(ys), np.concatenate(it),
        np.array(bw), np.array(bh), np.array(mn), np.array(mx),
        np.array(smin) if has_slide else None, np.array(smax) if has_slide else None)
