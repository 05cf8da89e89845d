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
