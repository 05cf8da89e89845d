"""GPU tests of the tile ABI v2 (nyxhip_featurize_tiles_v2 / _sharded): arbitrary label values, native element types, device
prescan, chunking by device budget, several contexts -- against host assembly (the restated phases 1-2) + the CPU oracle."""
import numpy as np
import pytest

import nyxus_amd
from nyxus_amd import _abi, _lib
from tests import roi_assembly
from oracle import pyoracle as po
from tests import parity, synth

pytestmark = pytest.mark.gpu
DBL_MAX = 1.7976931348623157e308
MASK = _abi.FAM_INTENSITY | _abi.FAM_GLCM


def _oracle_stack(I, M, mask, s, slide="montage"):
    tiles, labels, rows = [], [], []
    for t in range(I.shape[0]):
        it, lb = I[t].astype(np.uint32), M[t].astype(np.uint32)
        if slide == "montage":
            smin, smax = DBL_MAX, -DBL_MAX
        else:
            fg = it[lb != 0]
            smin, smax = (float(fg.min()), float(fg.max())) if fg.size else (0.0, 0.0)
        b = roi_assembly.assemble(it, lb, smin, smax)
        if b is None:
            continue
        rows.append(po.oracle_featurize(b, mask, s))
        labels.append(b.roi_label)
        tiles.append(np.full(len(b.roi_label), t, np.uint32))
    return np.concatenate(tiles), np.concatenate(labels), np.concatenate(rows)


def _blocks(rng, shape, labels):
    """A label tile with one rectangle per label value (some touching the border), random order of placement."""
    h, w = shape
    lab = np.zeros(shape, np.uint32)
    for k, l in enumerate(labels):
        y0 = (k * 11) % (h - 9); x0 = (k * 17) % (w - 12)
        lab[y0:y0 + 5 + k % 4, x0:x0 + 6 + k % 6] = l
    return lab


def test_arbitrary_label_values_one_tile_and_stack(hip_ctx):
    rng = np.random.default_rng(11)
    s = _abi.default_settings(8)
    big = [1, 10**6, 2**32 - 1, 7, 3_000_000_000, 65536]
    I = rng.integers(1, 4096, (3, 64, 80)).astype(np.uint32)
    M = np.stack([_blocks(rng, (64, 80), big), _blocks(rng, (64, 80), big[::-1]), np.zeros((64, 80), np.uint32)])
    M[2, 10:20, 10:30] = 2**31                                     # third tile: one ROI; a tile may also be empty (see below)
    for stack_i, stack_m in ((I[:1], M[:1]), (I, M)):
        tiles, labels, T = hip_ctx.featurize_tiles_host(stack_i, stack_m, MASK, s)
        wt, wl, want = _oracle_stack(stack_i, stack_m, MASK, s)
        assert tiles.tolist() == wt.tolist() and labels.tolist() == wl.tolist()      # (tile, label) ascending, values intact
        assert not parity.compare_tables(T, want, _lib.column_names(MASK, s))
    tiles, labels, T = hip_ctx.featurize_tiles_host(I[:2] * 0 + 5, M[:2] * 0, MASK, s)
    assert len(labels) == 0 and T.shape == (0, len(_lib.column_names(MASK, s)))


def test_native_element_types_equal_uint32(hip_ctx):
    rng = np.random.default_rng(12)
    s = _abi.default_settings(8)
    lab = synth.disk_label_tile(size=256, pitch=36, radius=14)       # 49 ROIs, labels < 256
    I = rng.integers(1, 250, (2, 256, 256))
    M = np.stack([lab, lab[::-1].copy()])
    ref = hip_ctx.featurize_tiles_host(I.astype(np.uint32), M.astype(np.uint32), MASK, s)
    for ti, tl in ((np.uint8, np.uint8), (np.uint16, np.uint8), (np.uint8, np.uint16), (np.uint16, np.uint32), (np.uint32, np.uint16)):
        got = hip_ctx.featurize_tiles_host(I.astype(ti), M.astype(tl), MASK, s)
        for a, b in zip(ref, got):
            assert np.array_equal(a, b, equal_nan=True), (ti, tl)     # bit-identical: only the load width differs


def test_device_prescan_matches_slide_props(hip_ctx):
    rng = np.random.default_rng(13)
    s = _abi.default_settings(16)
    mask = _abi.FAM_INTENSITY
    I = rng.integers(0, 3000, (2, 96, 96)).astype(np.uint16)
    M = np.stack([_blocks(rng, (96, 96), [5, 9, 300]), _blocks(rng, (96, 96), [2, 4])]).astype(np.uint16)
    tiles, labels, T = hip_ctx.featurize_tiles_host(I, M, mask, s, slide_mode=_abi.SLIDE_PER_TILE)
    wt, wl, want = _oracle_stack(I, M, mask, s, slide="per_tile")
    assert labels.tolist() == wl.tolist()
    names = _lib.column_names(mask, s)
    assert not parity.compare_tables(T, want, names)
    j = names.index("COVERED_IMAGE_INTENSITY_RANGE")
    assert np.all(T[:, j] > 0) and np.array_equal(T[:, j], want[:, j])
    smin, smax = np.array([10.0, 0.0]), np.array([5000.0, 6000.0])     # caller-given extrema (a slide cut into tiles)
    _, _, Tg = hip_ctx.featurize_tiles_host(I, M, mask, s, slide_mode=_abi.SLIDE_GIVEN, slide_min=smin, slide_max=smax)
    rng_col = T[:, names.index("RANGE")]
    assert np.array_equal(Tg[:, j], rng_col / (smax - smin)[tiles])


def test_chunking_by_device_budget_is_invisible(hip_ctx):
    rng = np.random.default_rng(14)
    s = _abi.default_settings(8)
    lab = synth.disk_label_tile(size=128, pitch=32, radius=12)
    I = rng.integers(1, 4096, (37, 128, 128)).astype(np.uint16)
    M = np.stack([np.roll(lab, k, axis=1) * (1 + k % 3) for k in range(37)]).astype(np.uint32)
    one = hip_ctx.featurize_tiles_host(I, M, MASK, s, max_device_bytes=1 << 34)
    many = hip_ctx.featurize_tiles_host(I, M, MASK, s, max_device_bytes=3 << 20)      # a few tiles per chunk
    for a, b in zip(one, many):
        assert np.array_equal(a, b, equal_nan=True)
    assert len(one[1]) == 37 * 16 and one[0].tolist() == sorted(one[0].tolist())


def test_two_contexts_share_a_stack(hip_ctx):
    rng = np.random.default_rng(15)
    s = _abi.default_settings(8)
    lab = synth.disk_label_tile(size=128, pitch=32, radius=12)
    I = rng.integers(1, 4096, (9, 128, 128)).astype(np.uint32)
    M = np.stack([lab * (k + 1) for k in range(9)]).astype(np.uint32)
    other = _lib.Context(0)                                        # a second context (same GPU here; one per GPU on a node)
    third = _lib.Context(0)
    try:
        one = hip_ctx.featurize_tiles_host(I, M, MASK, s)
        for extra in ([other], [other, third]):
            shared = hip_ctx.featurize_tiles_host(I, M, MASK, s, contexts=extra)
            for a, b in zip(one, shared):
                assert np.array_equal(a, b, equal_nan=True)
        few = hip_ctx.featurize_tiles_host(I[:1], M[:1], MASK, s, contexts=[other, third])   # fewer tiles than contexts
        assert np.array_equal(few[2], one[2][: len(few[1])], equal_nan=True)
    finally:
        other.close(); third.close()


def _varied_stack(seed=16, nt=12):
    """Tiles whose ROI sizes differ from tile to tile (radius 4 .. 26, one tile with a single 120-px-wide blob): chunks and shards of
    this stack have different extrema, hence different carve-outs and size classes."""
    rng = np.random.default_rng(seed)
    I = rng.integers(0, 4096, (nt, 128, 128)).astype(np.uint16)
    M = np.zeros((nt, 128, 128), np.uint32)
    for k in range(nt):
        if k == nt // 2:
            yy, xx = np.mgrid[0:128, 0:128]
            M[k][(xx - 64) ** 2 + (yy - 64) ** 2 <= 60 * 60] = 7
        else:
            r = 4 + (k * 7) % 23
            M[k] = synth.disk_label_tile(size=128, pitch=max(2 * r + 3, 16), radius=r) * (1 + k % 3)
    return I, M


def test_chunking_and_sharding_are_invisible_for_every_family(hip_ctx):
    """ADVICE r2 (GLSZM bits followed the launch extrema): with ROI sizes that differ between tiles, the rows of every family
    -- texture, dependence, shape and moment families included -- are the same bits whatever the chunk size or context count."""
    I, M = _varied_stack()
    s = _abi.default_settings(8)
    mask = _abi.FAM_ALL
    one = hip_ctx.featurize_tiles_host(I, M, mask, s, max_device_bytes=1 << 34)
    many = hip_ctx.featurize_tiles_host(I, M, mask, s, max_device_bytes=3 << 20)
    other = _lib.Context(0)
    try:
        shared = hip_ctx.featurize_tiles_host(I, M, mask, s, contexts=[other])
    finally:
        other.close()
    names = _lib.column_names(mask, s)
    for tag, alt in (("chunked", many), ("two contexts", shared)):
        assert np.array_equal(one[0], alt[0]) and np.array_equal(one[1], alt[1])
        ne = ~((one[2] == alt[2]) | (np.isnan(one[2]) & np.isnan(alt[2])))
        assert not ne.any(), (tag, sorted({names[j] for j in np.nonzero(ne)[1]})[:12])


def test_label_confetti_grows_the_tile_table(hip_ctx):
    rng = np.random.default_rng(16)
    s = _abi.default_settings(8)
    mask = _abi.FAM_INTENSITY
    M = np.zeros((1, 64, 64), np.uint32)
    ys, xs = np.mgrid[0:64:2, 0:64:2]
    M[0, ys, xs] = rng.permutation(np.arange(1, 1025)).reshape(32, 32) * 4099      # 1024 single-pixel ROIs, sparse values
    I = rng.integers(1, 100, (1, 64, 64)).astype(np.uint32)
    tiles, labels, T = hip_ctx.featurize_tiles_host(I, M, mask, s)
    wt, wl, want = _oracle_stack(I, M, mask, s)
    assert labels.tolist() == wl.tolist() and len(labels) == 1024
    assert not parity.compare_tables(T, want, _lib.column_names(mask, s))


def test_two_thousand_tiles_in_one_call(hip_ctx):
    s = _abi.default_settings(8)
    lab = synth.disk_label_tile(size=128, pitch=64, radius=20)       # 4 ROIs per tile
    rng = np.random.default_rng(17)
    I = rng.integers(1, 4096, (2000, 128, 128)).astype(np.uint16)
    M = np.broadcast_to(lab.astype(np.uint8), (2000, 128, 128))
    tiles, labels, T = hip_ctx.featurize_tiles_host(I, np.ascontiguousarray(M), MASK, s, max_device_bytes=64 << 20)
    assert len(labels) == 8000 and tiles[-1] == 1999 and np.isfinite(T[:, 0]).all()
    wt, wl, want = _oracle_stack(I[1990:1991], np.ascontiguousarray(M[1990:1991]), MASK, s)
    assert not parity.compare_tables(T[tiles == 1990], want, _lib.column_names(MASK, s))


def test_nyxus_gpu_devices_and_ram_limit():
    rng = np.random.default_rng(18)
    lab = synth.disk_label_tile(size=128, pitch=32, radius=12)
    I = rng.integers(1, 4096, (6, 128, 128)).astype(np.uint16)
    M = np.stack([lab] * 6).astype(np.uint16)
    a = nyxus_amd.Nyxus(["*ALL_INTENSITY*", "*ALL_GLCM*"], coarse_gray_depth=8).featurize(I, M)
    b = nyxus_amd.Nyxus(["*ALL_INTENSITY*", "*ALL_GLCM*"], coarse_gray_depth=8, gpu_devices=[0, 0], ram_limit=8).featurize(I, M)
    assert list(a.columns) == list(b.columns) and a.shape == b.shape == (96, 4 + 36 + 149)
    assert a.equals(b)


@pytest.mark.parametrize("gd", [8, 64])
def test_window_mode_is_bit_identical_to_cloud_mode(hip_ctx, gd):
    """INTENSITY / GLCM alone read the ROIs' windows of the tiles inside the feature kernel; any other family makes the tile path
    materialise clouds first.  Both must give the same bits for the shared columns (a pixel's index in the kernel's value
    buffer is its rank in window order either way)."""
    rng = np.random.default_rng(21)
    s = _abi.default_settings(gd)
    h, w = 200, 260
    M = np.zeros((3, h, w), np.uint16)
    for t in range(3):
        for k in range(1, 40):                                   # random overlapping blobs: concave ROIs, holes, border contact
            cy, cx, r = rng.integers(0, h), rng.integers(0, w), rng.integers(2, 30)
            yy, xx = np.ogrid[:h, :w]
            M[t][((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r) & (rng.random((h, w)) > 0.05)] = k * 7
    I = rng.integers(0, 3000, (3, h, w)).astype(np.uint16)
    I[0][M[0] == 7] = 55                                          # a constant ROI
    I[1][M[1] == 14] = 0                                          # a blank ROI
    t1, l1, T1 = hip_ctx.featurize_tiles_host(I, M, MASK, s)                         # window mode
    t2, l2, T2 = hip_ctx.featurize_tiles_host(I, M, MASK | _abi.FAM_NGTDM, s)        # cloud mode (NGTDM needs clouds)
    assert np.array_equal(t1, t2) and np.array_equal(l1, l2) and len(l1) > 60
    n1 = _lib.column_names(MASK, s)
    n2 = _lib.column_names(MASK | _abi.FAM_NGTDM, s)
    sel = [n2.index(c) for c in n1]
    assert np.array_equal(T1, T2[:, sel], equal_nan=True)
    wt, wl, want = _oracle_stack(I, M, MASK, s)
    assert not parity.compare_tables(T1, want, n1)


@pytest.mark.parametrize("hi", [200, 256, 1001])
def test_tile_path_under_ibsi_with_levels_beyond_lds(hip_ctx, hi):
    """ibsi=True through the tile path (round-3 advisor's first trigger, never exercised): the co-occurrence matrix is as large as
    the ROI's largest intensity, so with intensities >= ~170 a window-mode chunk meets a class that needs the workspace -- the
    chunk materialises its clouds and runs again.  Rows equal host assembly + oracle."""
    rng = np.random.default_rng(51)
    s = _abi.default_settings(8)
    s.ibsi = 1
    lab = synth.disk_label_tile(size=192, pitch=48, radius=20)
    I = rng.integers(1, hi, (2, 192, 192)).astype(np.uint32)
    M = np.stack([lab, lab[::-1].copy()]).astype(np.uint32)
    for mask in (MASK, MASK | _abi.FAM_GLRLM | _abi.FAM_NGTDM):
        tiles, labels, T = hip_ctx.featurize_tiles_host(I, M, mask, s)
        wt, wl, want = _oracle_stack(I, M, mask, s)
        assert tiles.tolist() == wt.tolist() and labels.tolist() == wl.tolist()
        bad = parity.compare_tables(T, want, _lib.column_names(mask, s))
        assert not bad, "\n".join(bad[:10])


def test_tile_path_with_a_large_wide_range_roi(hip_ctx):
    """A 25 k-pixel ROI whose uint32 intensities span more than 16 bits, next to small ROIs, through the tile path: the chunk falls
    back to clouds (the ROI is beyond the LDS classes) and the histogram / sort engines behind it see the full range."""
    rng = np.random.default_rng(52)
    H = W = 384
    yy, xx = np.mgrid[0:H, 0:W]
    lab = np.zeros((H, W), np.uint32)
    lab[(yy - 150) ** 2 + (xx - 150) ** 2 <= 90 ** 2] = 7            # 25.4 k pixels
    lab[(yy - 330) ** 2 + (xx - 330) ** 2 <= 25 ** 2] = 3
    lab[(yy - 40) ** 2 + (xx - 330) ** 2 <= 12 ** 2] = 11
    I = rng.integers(1, 4096, (1, H, W)).astype(np.uint32)
    big = lab == 7
    I[0][big] = rng.integers(5, 100_000, int(big.sum())).astype(np.uint32)
    I[0][150, 150] = 3_000_000                                        # one hot pixel: a range of 3e6, beyond every 16-bit table
    s = _abi.default_settings(8)
    tiles, labels, T = hip_ctx.featurize_tiles_host(I, lab[None], MASK, s)
    wt, wl, want = _oracle_stack(I, lab[None], MASK, s)
    assert labels.tolist() == wl.tolist() == [3, 7, 11]
    bad = parity.compare_tables(T, want, _lib.column_names(MASK, s))
    assert not bad, "\n".join(bad[:10])
