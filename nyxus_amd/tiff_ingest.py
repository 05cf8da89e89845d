"""TIFF tile ingest for `featurize_directory` (a "next" row).

The reference reads strip / tile (OME-)TIFF through libtiff, tile by tile, into uint32 tile buffers
(/root/reference/src/nyx/image_loader.cpp, grayscale_tiff.h:104-200 tiles, :473-560 strips).  Here the same decode -- tile by
tile or strip by strip, libtiff underneath -- is the native reader of include/nyxtiff.h (nyxus_amd/csrc/tiff_reader.cpp ->
libnyxtiff.so); the image lands in a NumPy array of the file's own unsigned width (8 / 16 / 32 bits), which the device path
takes as it is.  Signed samples: negatives clamp to 0; floating point: refused (both as documented in the header).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnyxtiff.so")
_lib = None


class _Info(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("bits_per_sample", C.c_uint32), ("sample_format", C.c_uint32),
                ("tile_width", C.c_uint32), ("tile_height", C.c_uint32), ("samples_per_pixel", C.c_uint32)]


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise ImportError(f"{_LIB_PATH} is missing: build it with `make -C nyxus_amd/csrc` (python -c 'import __graft_entry__ as g; g.build()')")
        lib = C.CDLL(_LIB_PATH)
        lib.nyxtiff_info.argtypes = [C.c_char_p, C.POINTER(_Info), C.c_char_p, C.c_size_t]
        lib.nyxtiff_info.restype = C.c_int
        lib.nyxtiff_read.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t]
        lib.nyxtiff_read.restype = C.c_int
        _lib = lib
    return _lib


def tiff_info(path: str) -> dict:
    lib = _load()
    info = _Info()
    err = C.create_string_buffer(512)
    if lib.nyxtiff_info(os.fsencode(path), C.byref(info), err, 512) != 0:
        raise IOError(err.value.decode() or f"cannot read {path}")
    return {k: getattr(info, k) for k, _ in _Info._fields_}


def read_tiff(path: str) -> np.ndarray:
    """First page of a tiled or stripped TIFF as a 2-D uint8 / uint16 / uint32 array (the file's own sample width)."""
    lib = _load()
    i = tiff_info(path)
    dt = {8: np.uint8, 16: np.uint16}.get(i["bits_per_sample"], np.uint32)
    a = np.empty((i["height"], i["width"]), dt)
    err = C.create_string_buffer(512)
    if lib.nyxtiff_read(os.fsencode(path), a.ctypes.data, a.dtype.itemsize, i["width"], i["height"], err, 512) != 0:
        msg = err.value.decode()
        raise ValueError(msg) if "floating-point" in msg else IOError(msg)
    return a
