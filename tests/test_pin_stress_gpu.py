"""Stress test of the host side of the tile path (nyxhip_api.hip: staged_h2d / HostPin).

Rounds 3-5 registered the caller's arrays (hipHostRegister) whenever /proc/self/maps made them look like mappings of their own;
pages of malloc arenas registered and released left the driver in a state in which a LATER copy faulted on the GPU ("Memory access
fault by GPU ... on address <a page of the host heap>", 5 of 48 and later 4 of 33 runs of the GPU suite).  Round 6: pageable memory
(NYXHIP_MEM_HOST) goes through the library's own pinned staging ring, and the caller's pages are registered only on the caller's
statement (NYXHIP_MEM_HOST_OWN_MAPPING: whole pages inside the array, rounded inward).  These tests drive the pattern that used to
fault -- hundreds of calls alternating small heap arrays, large arrays and unaligned slices of larger arrays -- through the staging
ring, and the same with mmap-backed stacks declared as such."""
import numpy as np
import pytest

from nyxus_amd import _abi

pytestmark = pytest.mark.gpu

MASK = _abi.FAM_INTENSITY | _abi.FAM_GLCM


def _labels(n, size, rng):
    lab = np.zeros((n, size, size), np.uint32)
    yy, xx = np.mgrid[0:size, 0:size]
    for t in range(n):
        for k in range(1, 5):
            cy, cx, r = rng.integers(8, size - 8, 2).tolist() + [int(rng.integers(3, 8))]
            lab[t][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = k
    return lab


def _mmap_array(shape, dtype):
    """A numpy array over an anonymous mapping of its own (page-aligned, never part of an allocator arena)."""
    import mmap
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    m = mmap.mmap(-1, max(n, 4096))
    return np.frombuffer(m, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


@pytest.mark.parametrize("mode", ["staged", "own_mapping"])
def test_host_path_survives_alternating_small_large_and_unaligned_arrays(hip_ctx, mode):
    rng = np.random.default_rng(23)
    s = _abi.default_settings(8)
    # small: two 64 x 64 tiles (32 KB per array: heap-allocated, neighbours share pages)
    sm_lab = _labels(2, 64, rng)
    sm_int = rng.integers(1, 4096, sm_lab.shape).astype(np.uint32)
    # large: three 1024 x 1024 tiles (12 MiB per array: a mapping of its own)
    lg_lab = _labels(3, 1024, rng)
    lg_int = rng.integers(1, 4096, lg_lab.shape).astype(np.uint32)
    # unaligned: a 9 MiB slice that starts 12 bytes into a larger array and ends short of it (edge pages shared with the rest)
    n_sl = 9 * 512 * 512
    big_i = rng.integers(1, 4096, n_sl + 1000).astype(np.uint32)
    big_l = np.zeros(n_sl + 1000, np.uint32)
    sl_lab = big_l[3:3 + n_sl].reshape(9, 512, 512)
    sl_lab[:] = _labels(9, 512, rng)
    sl_int = big_i[3:3 + n_sl].reshape(9, 512, 512)
    assert sl_int.ctypes.data % 4096 != 0 and sl_lab.ctypes.data % 4096 != 0
    own = [False, False, False]
    if mode == "own_mapping":                       # the large stacks again as mappings of their own, declared; the unaligned slices of one too
        lg_lab2 = _mmap_array(lg_lab.shape, np.uint32); lg_lab2[:] = lg_lab
        lg_int2 = _mmap_array(lg_int.shape, np.uint32); lg_int2[:] = lg_int
        lg_lab, lg_int = lg_lab2, lg_int2
        big_l2 = _mmap_array(big_l.shape, np.uint32); big_l2[:] = big_l
        big_i2 = _mmap_array(big_i.shape, np.uint32); big_i2[:] = big_i
        sl_lab = big_l2[3:3 + n_sl].reshape(9, 512, 512); sl_int = big_i2[3:3 + n_sl].reshape(9, 512, 512)
        own = [False, True, True]
    cases = [(sm_int, sm_lab), (lg_int, lg_lab), (sl_int, sl_lab)]
    first = [None, None, None]
    churn = []
    for it in range(300):
        k = it % 3
        # fresh small arrays now and then: the allocator hands their pages out again and again
        if k == 0 and it % 9 == 0:
            churn = [rng.integers(1, 4096, sm_lab.shape).astype(np.uint32) for _ in range(3)]
            inten, lab = churn[it % 3], sm_lab.copy()
            _t, labels, table = hip_ctx.featurize_tiles_host(inten, lab, MASK, s)
            assert len(labels) == 8 and np.isfinite(table[:, 0]).all()
            continue
        inten, lab = cases[k]
        _t, labels, table = hip_ctx.featurize_tiles_host(inten, lab, MASK, s, own_mapping=own[k])
        if first[k] is None:
            first[k] = (labels.copy(), table.copy())
        else:
            assert np.array_equal(labels, first[k][0])
            assert np.array_equal(table, first[k][1], equal_nan=True)


@pytest.mark.parametrize("arenas", ["1", "64"])
def test_host_path_under_other_arena_settings(arenas):
    """The host path no longer looks at the allocator at all; MALLOC_ARENA_MAX changes where the arrays land (1: everything in the
    program-break heap or mappings of its own; 64: up to 64 arenas with threads).  A fresh process per setting (the variable is read
    when the allocator starts) runs a short version of the stress loop, a thread allocating beside it."""
    import os
    import subprocess
    import sys
    code = r"""
import threading, numpy as np
from nyxus_amd import _abi, _lib
ctx = _lib.Context(0)
rng = np.random.default_rng(5)
s = _abi.default_settings(8)
mask = _abi.FAM_INTENSITY | _abi.FAM_GLCM
stop = False
def churn():
    keep = []
    while not stop:
        keep.append(np.ones(int(rng.integers(1, 3_000_000)), np.uint8))
        if len(keep) > 8: keep.pop(0)
th = threading.Thread(target=churn); th.start()
first = {}
try:
    for it in range(60):
        n, size = ((2, 64), (3, 1024), (9, 512))[it % 3]
        lab = np.zeros((n, size, size), np.uint32); lab[:, 8:30, 8:40] = 1; lab[:, size - 30:size - 5, 5:25] = 2
        inten = np.random.default_rng(it % 3).integers(1, 4096, lab.shape).astype(np.uint32)
        _t, labels, table = ctx.featurize_tiles_host(inten, lab, mask, s)
        assert len(labels) == 2 * n and np.isfinite(table[:, 0]).all()
        k = it % 3
        if k in first: assert np.array_equal(table, first[k], equal_nan=True)
        else: first[k] = table.copy()
finally:
    stop = True; th.join()
ctx.close()
print("ok")
"""
    env = dict(os.environ, MALLOC_ARENA_MAX=arenas)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
