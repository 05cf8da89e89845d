"""Loader and thin Python face of the C ABI (``include/nyxhip.h``).

The HIP library is the product: if ``libnyxhip.so`` is missing or no GPU is
present this module raises -- there is no CPU fallback on the product path.
"""
from __future__ import annotations

import atexit
import ctypes as C
import os
import sys
import weakref
from typing import List, Optional

import numpy as np

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# NYXHIP_LIB selects another build of the same ABI (e.g. the stamped diagnostic build)
LIB_PATH = os.environ.get("NYXHIP_LIB") or os.path.join(_HERE, "libnyxhip.so")

# every symbol include/nyxhip.h declares
ABI_SYMBOLS = [
    "nyxhip_abi_version", "nyxhip_default_settings", "nyxhip_init", "nyxhip_destroy", "nyxhip_last_error",
    "nyxhip_set_stream", "nyxhip_n_columns", "nyxhip_column_name", "nyxhip_featurize_batch",
    "nyxhip_featurize_batch_async", "nyxhip_sync", "nyxhip_finalize_table", "nyxhip_featurize_tile", "nyxhip_featurize_tiles",
    "nyxhip_timing_enable", "nyxhip_timing_reset", "nyxhip_timing_get",
    "nyxhip_featurize_tiles_v2", "nyxhip_fetch_result", "nyxhip_featurize_tiles_sharded", "nyxhip_fetch_result_sharded",
    "nyxhip_launch_report",
]


class NyxHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"nyxhip error {code} ({_abi.ERR_NAMES.get(code, '?')}): {msg}")
        self.code = code


_lib = None


def load() -> C.CDLL:
    """dlopen()s libnyxhip.so and declares the prototypes.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C nyxus_amd/csrc).  The MI355X path has no CPU fallback.")
    # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64 (same SONAME as
    # /opt/rocm's).  If libnyxhip.so pulled in the system copy first and torch loaded its own later,
    # the second runtime would see no GPU.  Importing torch first makes the dynamic linker resolve
    # libnyxhip.so against the already-loaded copy.  (Pure C/C++ users of the ABI are unaffected.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    lib.nyxhip_abi_version.restype = C.c_int
    lib.nyxhip_default_settings.argtypes = [P(_abi.Settings)]
    lib.nyxhip_default_settings.restype = None
    lib.nyxhip_init.argtypes = [C.c_int, P(C.c_void_p)]
    lib.nyxhip_init.restype = C.c_int
    lib.nyxhip_destroy.argtypes = [C.c_void_p]
    lib.nyxhip_destroy.restype = None
    lib.nyxhip_last_error.argtypes = [C.c_void_p]
    lib.nyxhip_last_error.restype = C.c_char_p
    lib.nyxhip_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.nyxhip_set_stream.restype = C.c_int
    lib.nyxhip_n_columns.argtypes = [C.c_uint32, P(_abi.Settings)]
    lib.nyxhip_n_columns.restype = C.c_int
    lib.nyxhip_column_name.argtypes = [C.c_uint32, P(_abi.Settings), C.c_int, C.c_char_p, C.c_size_t]
    lib.nyxhip_column_name.restype = C.c_int
    for name in ("nyxhip_featurize_batch", "nyxhip_featurize_batch_async"):
        f = getattr(lib, name)
        f.argtypes = [C.c_void_p, P(_abi.Batch), C.c_uint32, P(_abi.Settings), C.c_void_p, C.c_size_t]
        f.restype = C.c_int
    lib.nyxhip_sync.argtypes = [C.c_void_p]
    lib.nyxhip_sync.restype = C.c_int
    lib.nyxhip_finalize_table.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double]
    lib.nyxhip_finalize_table.restype = None
    lib.nyxhip_featurize_tile.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32,
                                          C.c_uint32, C.c_uint32, P(_abi.Settings), C.c_void_p, C.c_uint64,
                                          C.c_void_p, C.c_size_t, P(C.c_uint64)]
    lib.nyxhip_featurize_tile.restype = C.c_int
    lib.nyxhip_featurize_tiles.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32,
                                           C.c_uint32, C.c_uint32, P(_abi.Settings), C.c_void_p, C.c_void_p, C.c_uint64,
                                           C.c_void_p, C.c_size_t, P(C.c_uint64)]
    lib.nyxhip_featurize_tiles.restype = C.c_int
    lib.nyxhip_featurize_tiles_v2.argtypes = [C.c_void_p, P(_abi.Tiles), C.c_uint32, P(_abi.Settings), C.c_void_p, C.c_void_p, C.c_uint64,
                                              C.c_void_p, C.c_size_t, P(C.c_uint64)]
    lib.nyxhip_featurize_tiles_v2.restype = C.c_int
    lib.nyxhip_fetch_result.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.nyxhip_fetch_result.restype = C.c_int
    lib.nyxhip_featurize_tiles_sharded.argtypes = [P(C.c_void_p), C.c_int, P(_abi.Tiles), C.c_uint32, P(_abi.Settings), C.c_void_p, C.c_void_p,
                                                   C.c_uint64, C.c_void_p, C.c_size_t, P(C.c_uint64)]
    lib.nyxhip_featurize_tiles_sharded.restype = C.c_int
    lib.nyxhip_fetch_result_sharded.argtypes = [P(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.nyxhip_fetch_result_sharded.restype = C.c_int
    lib.nyxhip_timing_enable.argtypes = [C.c_void_p, C.c_int]
    lib.nyxhip_timing_enable.restype = C.c_int
    lib.nyxhip_timing_reset.argtypes = [C.c_void_p]
    lib.nyxhip_timing_reset.restype = C.c_int
    lib.nyxhip_timing_get.argtypes = [C.c_void_p, P(C.c_double), P(C.c_uint64)]
    lib.nyxhip_timing_get.restype = C.c_int
    if hasattr(lib, "nyxhip_launch_report"):     # (absent from older builds of the ABI selected through NYXHIP_LIB for A/B runs)
        lib.nyxhip_launch_report.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        lib.nyxhip_launch_report.restype = C.c_int
    _lib = lib
    return lib


def column_names(mask: int, s: _abi.Settings) -> List[str]:
    lib = load()
    n = lib.nyxhip_n_columns(mask, C.byref(s))
    buf = C.create_string_buffer(128)
    out = []
    for i in range(n):
        rc = lib.nyxhip_column_name(mask, C.byref(s), i, buf, 128)
        if rc != 0:
            raise NyxHipError(rc, f"column {i}")
        out.append(buf.value.decode())
    return out


# Contexts still open at interpreter exit are destroyed here, while the HIP runtime is certainly alive (Python's atexit handlers
# run before the C library's): a context freed by the garbage collector during shutdown could reach hipFree / hipStreamDestroy
# after the runtime's own exit handlers.
_live_contexts = weakref.WeakSet()


def _close_live_contexts():
    for ctx in list(_live_contexts):
        try:
            ctx.close()
        except Exception:
            pass


atexit.register(_close_live_contexts)


class Context:
    """One ``nyxhip_ctx`` bound to one GPU (one per process rank)."""

    def __init__(self, device: int = 0):
        self._lib = load()
        h = C.c_void_p()
        rc = self._lib.nyxhip_init(device, C.byref(h))
        if rc != 0:
            raise NyxHipError(rc, (self._lib.nyxhip_last_error(None) or b"").decode())
        self._h = h
        self.device = device
        _live_contexts.add(self)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.nyxhip_destroy(self._h)
            self._h = None

    def __del__(self):
        # a context that is still alive when the interpreter shuts down was closed by _close_live_contexts (atexit: before the
        # HIP runtime's own exit handlers); one collected later than that must not call into a runtime that may be gone
        try:
            if sys is None or sys.is_finalizing():    # (module globals may already be cleared at that point)
                return
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise NyxHipError(rc, (self._lib.nyxhip_last_error(self._h) or b"").decode())

    def set_stream(self, stream_ptr: Optional[int]):
        self._check(self._lib.nyxhip_set_stream(self._h, C.c_void_p(stream_ptr)))

    def n_columns(self, mask: int, s: _abi.Settings) -> int:
        return self._lib.nyxhip_n_columns(mask, C.byref(s))

    def featurize_host(self, batch: _abi.HostBatch, mask: int, s: _abi.Settings) -> np.ndarray:
        """Host arrays in, host table out (synchronous)."""
        ncol = self.n_columns(mask, s)
        out = np.empty((batch.n_roi, ncol), np.float64)
        cb = batch.c_struct()
        self._check(self._lib.nyxhip_featurize_batch(self._h, C.byref(cb), mask, C.byref(s), out.ctypes.data, ncol))
        return out

    def featurize_device_async(self, cb: _abi.Batch, mask: int, s: _abi.Settings, out_ptr: int, ld: int):
        """Device pointers in ``cb``; enqueues the kernel on the context's stream."""
        self._check(self._lib.nyxhip_featurize_batch_async(self._h, C.byref(cb), mask, C.byref(s), C.c_void_p(out_ptr), ld))

    def featurize_tile_host(self, inten: np.ndarray, label: np.ndarray, mask: int, s: _abi.Settings, max_label: Optional[int] = None):
        """One intensity / label tile pair (host arrays) through the fused device path: label scan + ROI assembly + reduce.
        Returns (labels ascending, table).  `max_label` (v1 semantics): a label above it is an error."""
        if inten.shape != label.shape or inten.ndim != 2:
            raise ValueError("tiles must be 2-D arrays of the same shape")
        if max_label is not None:
            inten = np.ascontiguousarray(inten, np.uint32)
            label = np.ascontiguousarray(label, np.uint32)
            ncol = self.n_columns(mask, s)
            cap = max(1, min(max_label, inten.size))
            labels = np.zeros(cap, np.uint32)
            table = np.empty((cap, ncol), np.float64)
            n = C.c_uint64(0)
            self._check(self._lib.nyxhip_featurize_tile(self._h, inten.ctypes.data, label.ctypes.data, inten.shape[1], inten.shape[0],
                                                        _abi.MEM_HOST, max_label, mask, C.byref(s), labels.ctypes.data, cap,
                                                        table.ctypes.data, ncol, C.byref(n)))
            return labels[: n.value], table[: n.value]
        _, labels, table = self.featurize_tiles_host(inten[None], label[None], mask, s)
        return labels, table

    def featurize_tiles_host(self, inten: np.ndarray, label: np.ndarray, mask: int, s: _abi.Settings, slide_mode: int = _abi.SLIDE_MONTAGE,
                             slide_min=None, slide_max=None, max_device_bytes: int = 0, contexts: Optional[list] = None, own_mapping: bool = False):
        """A stack [n_tiles, H, W] of host tiles through the fused device path (nyxhip_featurize_tiles_v2).  uint8 / uint16 /
        uint32 arrays are handed over as they are (the kernels widen); anything else is cast to uint32 first.  Label values are
        arbitrary.  Returns (tile_index, labels, table) with rows ordered by (tile, label), sized by the ROI count the device
        scan found.  `contexts`: more contexts (one per GPU) -> nyxhip_featurize_tiles_sharded block-partitions the stack.
        `own_mapping`: the caller's statement that both stacks are mappings of their own (e.g. numpy arrays over an mmap): they
        are registered for DMA in place (NYXHIP_MEM_HOST_OWN_MAPPING); otherwise the bytes pass through the library's pinned
        staging ring, which assumes nothing about the allocator."""
        if inten.shape != label.shape or inten.ndim != 3:
            raise ValueError("stacks must be 3-D arrays [n_tiles, H, W] of the same shape")
        dt = {np.dtype(np.uint8): _abi.U8, np.dtype(np.uint16): _abi.U16, np.dtype(np.uint32): _abi.U32}
        inten = np.ascontiguousarray(inten if inten.dtype in dt else inten.astype(np.uint32))
        label = np.ascontiguousarray(label if label.dtype in dt else label.astype(np.uint32))
        ncol = self.n_columns(mask, s)
        nt, h, w = inten.shape
        t = _abi.Tiles()
        t.inten = inten.ctypes.data; t.label = label.ctypes.data
        t.inten_dtype = dt[inten.dtype]; t.label_dtype = dt[label.dtype]
        t.width = w; t.height = h; t.n_tiles = nt; t.memory = _abi.MEM_HOST_OWN_MAPPING if own_mapping else _abi.MEM_HOST; t.slide_mode = slide_mode
        keep = []
        if slide_mode == _abi.SLIDE_GIVEN:
            keep = [np.ascontiguousarray(slide_min, np.float64), np.ascontiguousarray(slide_max, np.float64)]
            if keep[0].shape != (nt,) or keep[1].shape != (nt,):
                raise ValueError("slide_min / slide_max must hold one value per tile")
            t.slide_min = keep[0].ctypes.data; t.slide_max = keep[1].ctypes.data
        t.max_device_bytes = int(max_device_bytes)
        n = C.c_uint64(0)
        if nt == 0:
            return np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros((0, ncol))
        ctxs = [self] + [c for c in (contexts or []) if c is not self]
        if len(ctxs) > 1:
            arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
            self._check(self._lib.nyxhip_featurize_tiles_sharded(arr, len(ctxs), C.byref(t), mask, C.byref(s), None, None, 0, None, 0, C.byref(n)))
            labels = np.zeros(n.value, np.uint32); tiles = np.zeros(n.value, np.uint32); table = np.empty((n.value, ncol), np.float64)
            self._check(self._lib.nyxhip_fetch_result_sharded(arr, len(ctxs), labels.ctypes.data, tiles.ctypes.data, table.ctypes.data, ncol))
            return tiles, labels, table
        self._check(self._lib.nyxhip_featurize_tiles_v2(self._h, C.byref(t), mask, C.byref(s), None, None, 0, None, 0, C.byref(n)))
        labels = np.zeros(n.value, np.uint32)
        tiles = np.zeros(n.value, np.uint32)
        table = np.empty((n.value, ncol), np.float64)
        self._check(self._lib.nyxhip_fetch_result(self._h, labels.ctypes.data, tiles.ctypes.data, table.ctypes.data, ncol))
        return tiles, labels, table

    def sync(self):
        self._check(self._lib.nyxhip_sync(self._h))

    def timing(self, on: bool, groups: bool = False):
        """on: two events around every call (timing_get); groups: also around every launch group (the "ms" of launch_report)."""
        self._check(self._lib.nyxhip_timing_enable(self._h, (2 if groups else 1) if on else 0))
        self._check(self._lib.nyxhip_timing_reset(self._h))

    def timing_reset(self):
        self._check(self._lib.nyxhip_timing_reset(self._h))

    def launch_report(self):
        """The size classes of the last featurize_batch call as launched (list of dicts; nyxhip_launch_report)."""
        import json
        buf = C.create_string_buffer(1 << 14)
        n = self._lib.nyxhip_launch_report(self._h, buf, len(buf))
        if n < 0:
            raise NyxHipError(-n, "launch report")
        return json.loads(buf.value.decode())

    def timing_get(self):
        ms = C.c_double()
        n = C.c_uint64()
        self._check(self._lib.nyxhip_timing_get(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value
