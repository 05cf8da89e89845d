"""GPU tests of the large-ROI path (nyxus_amd/csrc/roi_large.hip): ROIs beyond the LDS size classes are cut into slabs and strips,
several workgroups per ROI, and everything the workgroups exchange is an integer added with atomics.

The reference gives any ROI to any worker thread (/root/reference/src/nyx/parallel.h:23-42); what these tests pin is that the cut is
invisible: parity with the oracle at 20 k / 100 k / 400 k pixels under every binning mode and table width, and rows that do not
depend on the companions, the workspace budget or the chunking."""
import os

import numpy as np
import pytest

from nyxus_amd import _abi, _lib
from oracle import pyoracle as po
from tests import parity
from tests.test_size_classes_gpu import ellipse_roi

pytestmark = pytest.mark.gpu

MASK = _abi.FAM_INTENSITY | _abi.FAM_GLCM
CONFIG4 = MASK | _abi.FAM_GLRLM | _abi.FAM_GLSZM | _abi.FAM_NGTDM


def large_rois(seed=3, hi=4096):
    """~20 k, ~100 k and ~400 k pixels, plus the awkward ones: zero intensities, holes, a constant ROI, an all-zero ROI, a thin
    2000 x 12 strip (size class 4 by its side alone), and a 17 k-pixel ROI just above the class boundary."""
    rng = np.random.default_rng(seed)
    rois = [ellipse_roi(90, 72, rng, hi=hi),                        # 20 k
            ellipse_roi(200, 160, rng, hi=hi, lo=0, holes=0.03),    # 100 k, zeros and holes
            ellipse_roi(400, 320, rng, hi=hi),                      # 400 k
            ellipse_roi(80, 68, rng, hi=hi)]                        # 17 k
    c = ellipse_roi(100, 70, rng)
    c["inten"][:] = 77
    rois.append(c)                                                  # constant
    z = ellipse_roi(95, 75, rng)
    z["inten"][:] = 0
    rois.append(z)                                                  # blank
    yy, xx = np.mgrid[0:12, 0:2000]
    o = np.lexsort((yy.ravel(), xx.ravel()))
    rois.append(dict(x=xx.ravel()[o], y=yy.ravel()[o], inten=rng.integers(1, hi, xx.size).astype(np.uint32)))
    return rois


def check(ctx, rois, mask, s, expect_coop=True):
    b = _abi.batch_from_rois(rois)
    G = ctx.featurize_host(b, mask, s)
    O = po.oracle_featurize(b, mask, s)
    bad = parity.compare_tables(G, O, _lib.column_names(mask, s), batch=b)
    assert not bad, "\n".join(bad[:20])
    rep = ctx.launch_report()
    if expect_coop:
        assert any(r["cooperative"] for r in rep), rep
    return G


@pytest.mark.parametrize("gd,ibsi", [(8, 0), (64, 0), (-16, 0), (100, 0), (8, 1)])
def test_large_rois_match_oracle(hip_ctx, gd, ibsi):
    s = _abi.default_settings(gd)
    s.ibsi = ibsi
    check(hip_ctx, large_rois(hi=200 if ibsi else 4096), MASK, s)


@pytest.mark.parametrize("hi", [256, 65536, 70000, 3000000])
def test_large_rois_of_every_table_width(hip_ctx, hi):
    """8-bit data (256 histogram entries shared by 100 k pixels), 16-bit data (the 128-KiB LDS table of the load kernel), ranges
    beyond 16 bits (global atomics) -- all below kLargeRangeMax."""
    check(hip_ctx, large_rois(seed=5, hi=hi)[:4], MASK, _abi.default_settings(8))


def test_ranges_beyond_the_histogram_take_the_sort_path(hip_ctx):
    rng = np.random.default_rng(7)
    rois = [ellipse_roi(90, 72, rng, hi=2 ** 31), ellipse_roi(100, 80, rng, hi=4096), ellipse_roi(85, 70, rng, hi=2 ** 23)]
    check(hip_ctx, rois, MASK, _abi.default_settings(8))


def test_intensity_alone_and_glcm_alone(hip_ctx):
    rois = large_rois(seed=9)[:3]
    s = _abi.default_settings(8)
    check(hip_ctx, rois, _abi.FAM_INTENSITY, s)
    check(hip_ctx, rois, _abi.FAM_GLCM, s)
    s2 = _abi.default_settings(8)
    s2.glcm_n_angles = 2
    s2.glcm_angles[0] = 45
    s2.glcm_angles[1] = 135
    s2.glcm_offset = 3
    s2.glcm_symmetric = 1
    check(hip_ctx, rois, _abi.FAM_GLCM, s2)


def test_config4_on_large_rois(hip_ctx):
    check(hip_ctx, large_rois(seed=11)[:3], CONFIG4, _abi.default_settings(8))
    assert any(r["cooperative"] & 2 for r in hip_ctx.launch_report())


@pytest.mark.parametrize("gd,ibsi", [(8, 0), (64, 0), (-16, 0), (100, 0), (300, 0), (8, 1)])
def test_texture_families_of_large_rois_match_oracle(hip_ctx, gd, ibsi):
    """GLRLM + GLSZM + NGTDM by several workgroups per ROI (roi_large_tex.hip) under every binning mode and plane width: the 20 k /
    100 k (zeros, holes) / 400 k-pixel ellipses, the constant and the all-zero ROI, the 2000 x 12 strip."""
    s = _abi.default_settings(gd)
    s.ibsi = ibsi
    tex = _abi.FAM_GLRLM | _abi.FAM_GLSZM | _abi.FAM_NGTDM
    check(hip_ctx, large_rois(seed=23, hi=200 if ibsi else 4096), tex, s, expect_coop=False)
    assert any(r["cooperative"] & 2 for r in hip_ctx.launch_report())


@pytest.mark.parametrize("fam", ["GLRLM", "GLSZM", "NGTDM"])
def test_each_texture_family_alone_on_large_rois(hip_ctx, fam):
    check(hip_ctx, large_rois(seed=29)[:4], getattr(_abi, "FAM_" + fam), _abi.default_settings(8), expect_coop=False)


def test_pin_holed_ellipse_of_400k_pixels(hip_ctx):
    """A 400 k-pixel ellipse with 2 % of its pixels missing (background inside the box everywhere: long zero runs end, zones of the
    background level split) and flat patches (long runs and zones that cross many strips), config-4 families."""
    rng = np.random.default_rng(31)
    r = ellipse_roi(400, 320, rng, holes=0.02)
    flat = (r["x"] // 37 + r["y"] // 29) % 5 == 0
    r["inten"][flat] = 2000
    check(hip_ctx, [r, ellipse_roi(120, 90, rng, holes=0.3, hi=300)], CONFIG4, _abi.default_settings(8))
    check(hip_ctx, [r], CONFIG4, _abi.default_settings(64))


def test_boxes_beyond_the_strip_path_fall_back(hip_ctx):
    """A 9000 x 20 box (wider than the strip kernels stage) next to boxes the path serves: the class is split between the two
    paths, every row matches."""
    rng = np.random.default_rng(37)
    yy, xx = np.mgrid[0:20, 0:9000]
    o = np.lexsort((yy.ravel(), xx.ravel()))
    wide = dict(x=xx.ravel()[o], y=yy.ravel()[o], inten=rng.integers(1, 4096, xx.size).astype(np.uint32))
    check(hip_ctx, [wide, ellipse_roi(300, 200, rng), ellipse_roi(90, 72, rng)], CONFIG4, _abi.default_settings(8))


def test_large_rows_do_not_depend_on_companions_budget_or_chunking(hip_ctx, monkeypatch):
    """20 k / 100 k / 400 k-pixel ROIs alone, among small and other large ROIs, and with a workspace budget that forces the class
    through chunks of one or two ROIs: equal bit for bit, every family the path touches and the ones beside it (Gabor included)."""
    rng = np.random.default_rng(13)
    big = [ellipse_roi(90, 72, rng), ellipse_roi(200, 160, rng, lo=0, holes=0.02), ellipse_roi(400, 320, rng), ellipse_roi(70, 60, rng, hi=60000)]
    small = [ellipse_roi(int(r), int(max(2, r - 2)), rng) for r in rng.integers(3, 40, 30)]
    others = [ellipse_roi(150, 150, rng, hi=300), ellipse_roi(260, 100, rng, hi=70000)]
    s = _abi.default_settings(8)
    mask = MASK | _abi.FAM_GLRLM | _abi.FAM_GLSZM | _abi.FAM_NGTDM | _abi.FAM_GABOR | _abi.FAM_ZERNIKE
    alone = hip_ctx.featurize_host(_abi.batch_from_rois(big), mask, s)
    pos = [0, 9, 17, 30]
    mixed_list = list(small) + others
    for p, r in zip(pos, big):
        mixed_list.insert(p, r)
    idx = [[i for i, q in enumerate(mixed_list) if q is r][0] for r in big]
    mixed = hip_ctx.featurize_host(_abi.batch_from_rois(mixed_list), mask, s)[idx]
    names = _lib.column_names(mask, s)

    def diff(a, c):
        return [(names[j], i) for i, j in zip(*np.nonzero(~((a == c) | (np.isnan(a) & np.isnan(c)))))]
    assert not diff(alone, mixed), diff(alone, mixed)[:10]
    monkeypatch.setenv("NYXHIP_LARGE_BUDGET_MB", "2")       # a 400 k-pixel ROI's block is ~0.7 MiB: chunks of a few ROIs
    tight = hip_ctx.featurize_host(_abi.batch_from_rois(mixed_list), mask, s)[idx]
    assert not diff(alone, tight), diff(alone, tight)[:10]
    monkeypatch.delenv("NYXHIP_LARGE_BUDGET_MB")


def test_many_large_rois_in_one_call(hip_ctx):
    """Sixty ROIs of 17 k .. 60 k pixels: the work maps hold hundreds of slabs and strips in arrival order; every row matches."""
    rng = np.random.default_rng(17)
    rois = [ellipse_roi(int(a), int(b), rng, hi=4096 if k % 3 else 40000) for k, (a, b) in enumerate(zip(rng.integers(80, 150, 60), rng.integers(70, 130, 60)))]
    check(hip_ctx, rois, MASK, _abi.default_settings(8))


def test_tile_path_with_a_large_roi_falls_back_to_clouds(hip_ctx):
    """A tile whose largest ROI is beyond the LDS classes (window mode does not apply to it): the chunk is served from clouds and
    equals the batch path."""
    rng = np.random.default_rng(19)
    H = W = 512
    lab = np.zeros((H, W), np.uint32)
    yy, xx = np.mgrid[0:H, 0:W]
    lab[(yy - 200) ** 2 + (xx - 200) ** 2 <= 150 ** 2] = 5             # 70 k pixels
    lab[(yy - 450) ** 2 + (xx - 450) ** 2 <= 30 ** 2] = 9
    lab[(yy - 60) ** 2 + (xx - 440) ** 2 <= 20 ** 2] = 2
    inten = rng.integers(1, 4096, (H, W)).astype(np.uint32)
    s = _abi.default_settings(8)
    _tiles, labels, T = hip_ctx.featurize_tiles_host(inten[None], lab[None], MASK, s)
    from tests import synth
    b = _abi.batch_from_rois(synth.rois_from_tile(inten, lab))
    O = po.oracle_featurize(b, MASK, s)
    assert list(labels) == [2, 5, 9]
    bad = parity.compare_tables(T, O, _lib.column_names(MASK, s), batch=b)
    assert not bad, "\n".join(bad[:20])


@pytest.mark.parametrize("gabor", [False, True])
def test_default_grey_depth_on_large_rois(hip_ctx, gabor):
    """The reference's default grey depth (64 levels: 64 x 64 matrices, 64 histogram bins) on 20 k / 100 k-pixel ROIs, every family
    of BASELINE configs[3], with and without Gabor beside them."""
    rng = np.random.default_rng(43)
    rois = [ellipse_roi(90, 72, rng), ellipse_roi(200, 160, rng, lo=0, holes=0.03), ellipse_roi(20, 18, rng)]
    mask = CONFIG4 | (_abi.FAM_GABOR if gabor else 0)
    check(hip_ctx, rois, mask, _abi.default_settings(64))


# ---- Gabor of ROIs beyond LDS by several workgroups per ROI (roi_large_gabor.hip, round 6) -------------------------------------------
def _same(G, O):
    return ((G == O) | (np.isnan(G) & np.isnan(O))).all()


def gabor_rois(seed=5, hi=4096):
    """Boxes on both sides of the 64 x 32 tiles of the strip path: 180 x 144 (20 k px), 401 x 321 with zeros and holes, a thin 2000 x 12
    strip, a 129 x 33 box, one tile plus one pixel each way (65 x 33 is LDS-sized: 200 x 130 instead), a constant and a blank ROI."""
    rng = np.random.default_rng(seed)
    rois = [ellipse_roi(90, 72, rng, hi=hi), ellipse_roi(200, 160, rng, hi=hi, lo=0, holes=0.03), ellipse_roi(100, 65, rng, hi=hi)]
    yy, xx = np.mgrid[0:12, 0:2000]
    o = np.lexsort((yy.ravel(), xx.ravel()))
    rois.append(dict(x=xx.ravel()[o], y=yy.ravel()[o], inten=rng.integers(1, hi, xx.size).astype(np.uint32)))
    yy, xx = np.mgrid[0:257, 0:129]
    v = (hi // 2 + (hi // 3) * np.sin(xx / 9.0) * np.cos(yy / 13.0)).astype(np.int64).clip(1, hi - 1)      # a smooth field: energies near the thresholds
    o = np.lexsort((yy.ravel(), xx.ravel()))
    rois.append(dict(x=xx.ravel()[o], y=yy.ravel()[o], inten=v.ravel()[o].astype(np.uint32)))
    c = ellipse_roi(100, 70, rng); c["inten"][:] = 77
    z = ellipse_roi(95, 75, rng); z["inten"][:] = 0
    return rois + [c, z]


@pytest.mark.parametrize("hi", [256, 4096, 65536, 1 << 25])
def test_gabor_of_large_rois_is_bit_identical(hip_ctx, hi):
    """Count ratios of every filter equal the oracle's bit for bit (default bank and an 8-filter bank), whatever the intensity depth:
    the strip path computes every energy with the reference's arithmetic."""
    b = _abi.batch_from_rois(gabor_rois(hi=hi))
    for nf in (4, 8):
        s = _abi.default_settings(8)
        if nf == 8:
            s.gabor_n_filters = 8
            for i in range(8):
                s.gabor_f0[i] = [4.0, 16.0, 32.0, 64.0][i % 4]
                s.gabor_theta[i] = float(np.pi * i / 8)
        G = hip_ctx.featurize_host(b, _abi.FAM_GABOR, s)
        assert any(r["cooperative"] & 4 for r in hip_ctx.launch_report()), hip_ctx.launch_report()
        assert _same(G, po.oracle_featurize(b, _abi.FAM_GABOR, s))


def test_gabor_strips_equal_the_one_workgroup_kernel(hip_ctx, monkeypatch):
    """Rows of large ROIs do not depend on the path (several workgroups per ROI or the one-workgroup workspace kernel), on companions
    or on the other families of the call."""
    rng = np.random.default_rng(8)
    big = gabor_rois(seed=8)
    small = [ellipse_roi(int(rng.integers(4, 30)), int(rng.integers(4, 30)), rng) for _ in range(40)]
    s = _abi.default_settings(8)
    alone = hip_ctx.featurize_host(_abi.batch_from_rois(big), _abi.FAM_GABOR, s)
    mixed = [small[i // 2] if i % 2 == 0 else big[i // 2] for i in range(2 * len(big))] + small[len(big):]
    idx_big = [i for i in range(2 * len(big)) if i % 2 == 1]
    bm = _abi.batch_from_rois(mixed)
    among = hip_ctx.featurize_host(bm, _abi.FAM_GABOR | _abi.FAM_ZERNIKE | _abi.FAM_INTENSITY, s)
    names = _lib.column_names(_abi.FAM_GABOR | _abi.FAM_ZERNIKE | _abi.FAM_INTENSITY, s)
    gcols = [i for i, n in enumerate(names) if n.startswith("GABOR")]
    assert _same(among[idx_big][:, gcols], alone)
    monkeypatch.setenv("NYXHIP_NO_COOP_GABOR", "1")
    one_wg = hip_ctx.featurize_host(_abi.batch_from_rois(big), _abi.FAM_GABOR, s)
    assert not any(r["cooperative"] & 4 for r in hip_ctx.launch_report())
    assert _same(one_wg, alone)


@pytest.mark.parametrize("n", [8, 16, 21])
def test_gabor_strips_with_other_kernel_sizes(hip_ctx, n):
    rng = np.random.default_rng(n)
    s = _abi.default_settings(8)
    s.gabor_kersize = n
    b = _abi.batch_from_rois([ellipse_roi(100, 80, rng), ellipse_roi(70, 90, rng, lo=0, holes=0.05)])
    assert _same(hip_ctx.featurize_host(b, _abi.FAM_GABOR, s), po.oracle_featurize(b, _abi.FAM_GABOR, s))
