"""TEST INFRASTRUCTURE -- ctypes access to the two CPU checkers:

* ``liboracle.so``        the plain-C restatement (oracle/nyx_oracle.c)
* ``_ref/libnyxref.so``   the real reference classes compiled in place (oracle/ref_driver.cpp)

Both take the same host ``nyxhip_batch`` the product ABI takes.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

from nyxus_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "liboracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libnyxref.so")


def build(quiet: bool = True) -> None:
    """(Re)builds liboracle.so and, when /root/reference exists, _ref/libnyxref.so."""
    subprocess.run(["make", "-C", _HERE, "-j8"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


_oracle = None
_ref = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        if not os.path.exists(_ORACLE_SO):
            build()
        lib = C.CDLL(_ORACLE_SO)
        lib.nyxo_n_columns.argtypes = [C.c_uint32, C.POINTER(_abi.Settings)]
        lib.nyxo_n_columns.restype = C.c_int
        lib.nyxo_featurize_batch.argtypes = [C.POINTER(_abi.Batch), C.c_uint32, C.POINTER(_abi.Settings),
                                             C.c_void_p, C.c_size_t]
        lib.nyxo_featurize_batch.restype = C.c_int
        _oracle = lib
    return _oracle


def have_ref() -> bool:
    return os.path.exists(_REF_SO)


def ref_lib():
    global _ref
    if _ref is None:
        lib = C.CDLL(_REF_SO)
        lib.nyxref_n_columns.argtypes = [C.c_uint32, C.POINTER(_abi.Settings)]
        lib.nyxref_n_columns.restype = C.c_int
        lib.nyxref_featurize_batch.argtypes = [C.POINTER(_abi.Batch), C.c_uint32, C.POINTER(_abi.Settings),
                                               C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
        lib.nyxref_featurize_batch.restype = C.c_int
        lib.nyxref_featurize_tiles.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_abi.Settings), C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        lib.nyxref_featurize_tiles.restype = C.c_int
        _ref = lib
    return _ref


def oracle_featurize(batch: _abi.HostBatch, mask: int, s: _abi.Settings) -> np.ndarray:
    lib = oracle_lib()
    ncol = lib.nyxo_n_columns(mask, C.byref(s))
    out = np.full((batch.n_roi, ncol), np.nan, np.float64)
    cb = batch.c_struct()
    rc = lib.nyxo_featurize_batch(C.byref(cb), mask, C.byref(s), out.ctypes.data, ncol)
    if rc != 0:
        raise RuntimeError(f"oracle: status {rc}")
    return out


def ref_featurize(batch: _abi.HostBatch, mask: int, s: _abi.Settings, n_threads: int = 1,
                  timing: Optional[list] = None) -> np.ndarray:
    lib = ref_lib()
    ncol = lib.nyxref_n_columns(mask, C.byref(s))
    out = np.full((batch.n_roi, ncol), np.nan, np.float64)
    cb = batch.c_struct()
    sec = C.c_double(0.0)
    rc = lib.nyxref_featurize_batch(C.byref(cb), mask, C.byref(s), out.ctypes.data, ncol, n_threads, C.byref(sec))
    if rc != 0:
        raise RuntimeError(f"reference driver: status {rc}")
    if timing is not None:
        timing.append(sec.value)
    return out


def ref_featurize_tiles(inten: np.ndarray, label: np.ndarray, mask: int, s: _abi.Settings, n_threads: int = 1, max_rows: Optional[int] = None,
                        timing: Optional[list] = None):
    """The reference's in-memory workflow on a stack [n_tiles, H, W] of uint32 tiles, scans included (oracle/ref_driver.cpp
    nyxref_featurize_tiles).  Returns (tile_index, labels, table); timing receives [scan_seconds, reduce_seconds]."""
    lib = ref_lib()
    inten = np.ascontiguousarray(inten, np.uint32)
    label = np.ascontiguousarray(label, np.uint32)
    nt, h, w = inten.shape
    ncol = lib.nyxref_n_columns(mask, C.byref(s))
    cap = int(max_rows) if max_rows is not None else 4096 * nt
    labels = np.zeros(cap, np.uint32)
    tiles = np.zeros(cap, np.uint32)
    out = np.full((cap, ncol), np.nan, np.float64)
    n = C.c_uint64(0)
    sec = (C.c_double * 2)()
    rc = lib.nyxref_featurize_tiles(inten.ctypes.data, label.ctypes.data, w, h, nt, mask, C.byref(s), n_threads, labels.ctypes.data, tiles.ctypes.data,
                                    out.ctypes.data, ncol, cap, C.byref(n), sec)
    if rc != 0:
        raise RuntimeError(f"reference tile driver: status {rc}")
    if n.value > cap:
        raise RuntimeError(f"reference tile driver: {n.value} rows, room for {cap}")
    if timing is not None:
        timing.extend([sec[0], sec[1]])
    return tiles[: n.value], labels[: n.value], out[: n.value]
