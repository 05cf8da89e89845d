"""TIFF tile ingest for `featurize_directory` (a "next" row).

The reference reads strip/tile (OME-)TIFF through libtiff into uint32 tile buffers
(/root/reference/src/nyx/image_loader.cpp, grayscale_tiff.h, raw_tiff.h).  File decoding is IO
plumbing outside the hot path; this round it goes through Pillow's libtiff binding and hands the kernel
path a 2-D integer array.  Multi-page / multi-channel files: first page, first channel (as the
reference's 2-D loader does for single-plane images).
"""
from __future__ import annotations

import numpy as np


def read_tiff(path: str) -> np.ndarray:
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None
    with Image.open(path) as im:
        im.seek(0)
        a = np.array(im)
    if a.ndim == 3:
        a = a[..., 0]
    if a.dtype.kind == "f":
        # float images: the reference rescales to an unsigned dynamic range (fpimage options); not on the hot path
        raise ValueError(f"{path}: floating-point TIFFs are outside the MI355X hot path (SURVEY.md section 8)")
    if a.dtype.kind == "i" and a.min() < 0:
        a = a - a.min()
    return np.ascontiguousarray(a)
