"""GPU tests of the 16-bit path (nyxus_amd/csrc/roi_wide.hip): LDS-sized ROIs whose intensity range rules the dense counting table out
get their first-order features from a presence bitmap + a duplicate list (no sort), the GLCM columns from the GLCM-only build.

Reference semantics pinned here: the exact columns (MEDIAN / MODE / percentiles / ROBUST_MEAN ...) of
/root/reference/src/nyx/features/histogram.h:214-309 on data with few, some and very many repeated values -- the last kind overflows
the duplicate list and takes the sort-based slow path of the same kernel."""
import numpy as np
import pytest

from nyxus_amd import _abi, _lib
from oracle import pyoracle as po
from tests import parity
from tests.test_size_classes_gpu import ellipse_roi

pytestmark = pytest.mark.gpu

MASK = _abi.FAM_INTENSITY | _abi.FAM_GLCM


def check(ctx, rois, mask, s):
    b = _abi.batch_from_rois(rois)
    G = ctx.featurize_host(b, mask, s)
    O = po.oracle_featurize(b, mask, s)
    bad = parity.compare_tables(G, O, _lib.column_names(mask, s), batch=b)
    assert not bad, "\n".join(bad[:20])
    return G


@pytest.mark.parametrize("gd", [8, 64, -16])
def test_sixteen_bit_rois_match_oracle(hip_ctx, gd):
    rng = np.random.default_rng(31)
    rois = [ellipse_roi(int(a), int(b), rng, hi=65536, lo=0 if k % 4 == 0 else 1, holes=0.1 if k % 3 == 0 else 0.0)
            for k, (a, b) in enumerate(zip(rng.integers(3, 60, 40), rng.integers(3, 60, 40)))]
    check(hip_ctx, rois, MASK, _abi.default_settings(gd))
    check(hip_ctx, rois[:10], _abi.FAM_INTENSITY, _abi.default_settings(gd))


def test_quantised_sixteen_bit_data_takes_the_slow_path(hip_ctx):
    """8-bit content in a 16-bit container (values k * 257): ~256 distinct values, each many times -- the duplicate list overflows."""
    rng = np.random.default_rng(32)
    rois = []
    for k in range(12):
        r = ellipse_roi(int(rng.integers(20, 60)), int(rng.integers(20, 60)), rng)
        r["inten"] = (rng.integers(0 if k % 2 else 1, 256, len(r["x"])) * 257).astype(np.uint32)
        rois.append(r)
    # two values only, far apart; a heavy mode; a range of exactly 65535 and of exactly 16384
    r = ellipse_roi(30, 30, rng); r["inten"] = np.where(rng.random(len(r["x"])) < 0.5, 5, 60005).astype(np.uint32); rois.append(r)
    r = ellipse_roi(25, 28, rng); v = rng.integers(100, 60000, len(r["x"])); v[rng.random(len(v)) < 0.3] = 31337; r["inten"] = v.astype(np.uint32); rois.append(r)
    r = ellipse_roi(20, 20, rng); v = rng.integers(0, 65536, len(r["x"])); v[0] = 0; v[1] = 65535; r["inten"] = v.astype(np.uint32); rois.append(r)
    r = ellipse_roi(20, 20, rng); v = rng.integers(7, 16392, len(r["x"])); v[0] = 7; v[1] = 16391; r["inten"] = v.astype(np.uint32); rois.append(r)
    check(hip_ctx, rois, MASK, _abi.default_settings(8))


def test_sixteen_bit_offsets_from_a_large_minimum(hip_ctx):
    """uint32 intensities around 3e9 with a 16-bit spread: the keys are offsets from the ROI minimum, the sums are not."""
    rng = np.random.default_rng(33)
    rois = []
    for k in range(8):
        r = ellipse_roi(int(rng.integers(10, 40)), int(rng.integers(10, 40)), rng)
        r["inten"] = (3_000_000_000 + rng.integers(0, 50000, len(r["x"]))).astype(np.uint32)
        rois.append(r)
    check(hip_ctx, rois, MASK, _abi.default_settings(8))


def test_sixteen_bit_tiles_through_the_tile_path(hip_ctx):
    """uint16 tiles: the wide classes of a window-mode chunk ask for the clouds; rows equal the batch path's."""
    from tests import synth
    rng = np.random.default_rng(34)
    lab = synth.disk_label_tile(size=256, pitch=64, radius=25).astype(np.uint16)
    inten = rng.integers(1, 65536, (2, 256, 256)).astype(np.uint16)
    s = _abi.default_settings(8)
    tiles, labels, T = hip_ctx.featurize_tiles_host(inten, np.stack([lab, lab]), MASK, s)
    rows = []
    for t in range(2):
        b = _abi.batch_from_rois(synth.rois_from_tile(inten[t].astype(np.uint32), lab.astype(np.uint32)))
        rows.append(po.oracle_featurize(b, MASK, s))
    O = np.concatenate(rows)
    assert len(labels) == len(O)
    bad = parity.compare_tables(T, O, _lib.column_names(MASK, s))
    assert not bad, "\n".join(bad[:20])


def test_glcm_alone_through_the_tile_path_and_the_batch_path(hip_ctx):
    """GLCM alone has a compile-time build of its own (roi_features_kernel_occ8<3, WIN>): the window loader (tile path) and the cloud
    loader (batch path) against the oracle, 8 and 16 levels, and a degenerate (constant) ROI among them."""
    from tests import synth
    rng = np.random.default_rng(41)
    lab = synth.disk_label_tile(size=256, pitch=64, radius=27).astype(np.uint32)
    inten = rng.integers(1, 4096, (3, 256, 256)).astype(np.uint32)
    inten[1][lab == 3] = 500                                               # a constant ROI: every GLCM column is the soft NaN
    for gd in (8, 16):
        s = _abi.default_settings(gd)
        tiles, labels, T = hip_ctx.featurize_tiles_host(inten, np.stack([lab] * 3), _abi.FAM_GLCM, s)
        rows = []
        for t in range(3):
            b = _abi.batch_from_rois(synth.rois_from_tile(inten[t], lab))
            rows.append(po.oracle_featurize(b, _abi.FAM_GLCM, s))
            G = hip_ctx.featurize_host(b, _abi.FAM_GLCM, s)
            assert not parity.compare_tables(G, rows[-1], _lib.column_names(_abi.FAM_GLCM, s))
        bad = parity.compare_tables(T, np.concatenate(rows), _lib.column_names(_abi.FAM_GLCM, s))
        assert not bad, "\n".join(bad[:20])


def test_gabor_on_intensities_beyond_fp32_integers(hip_ctx):
    """The Gabor screening pass runs in fp32, which holds integers below 2^24 exactly: an ROI with larger intensities takes the
    reference's arithmetic for every filter (exact columns), next to ordinary ROIs in the same launch."""
    rng = np.random.default_rng(42)
    rois = []
    for k in range(12):
        r = ellipse_roi(int(rng.integers(6, 30)), int(rng.integers(6, 30)), rng)
        hi = 2 ** 26 if k % 3 == 0 else 4096
        r["inten"] = rng.integers(1, hi, len(r["x"])).astype(np.uint32)
        rois.append(r)
    s = _abi.default_settings(8)
    b = _abi.batch_from_rois(rois)
    G = hip_ctx.featurize_host(b, _abi.FAM_GABOR, s)
    O = po.oracle_featurize(b, _abi.FAM_GABOR, s)
    assert np.array_equal(G, O, equal_nan=True), np.abs(G - O).max()


@pytest.mark.parametrize("gd", [8, 64])
def test_a_sixteen_bit_row_does_not_depend_on_wider_companions(hip_ctx, gd):
    """The engine that serves a wide-range ROI follows ITS range (<= 0xFFFF: bitmap path + GLCM-only build; beyond: the fused sort
    kernel), not the extrema of the class it shares (round-4 advisor, medium): rows of 16-bit ROIs alone, and next to 2^31-range
    ROIs of the same size classes, are equal bit for bit; so are the wide ones' rows; all of them match the oracle."""
    rng = np.random.default_rng(41)
    s = _abi.default_settings(gd)
    narrow = [ellipse_roi(7, 6, rng, hi=60000), ellipse_roi(30, 25, rng, hi=65536, lo=0), ellipse_roi(60, 50, rng, hi=40000)]
    wide = [ellipse_roi(8, 7, rng, hi=2 ** 31), ellipse_roi(28, 27, rng, hi=2 ** 31), ellipse_roi(20, 31, rng, hi=70000)]
    names = _lib.column_names(MASK, s)

    def diff(a, c):
        return [(names[j], i) for i, j in zip(*np.nonzero(~((a == c) | (np.isnan(a) & np.isnan(c)))))]
    alone_n = hip_ctx.featurize_host(_abi.batch_from_rois(narrow), MASK, s)
    alone_w = hip_ctx.featurize_host(_abi.batch_from_rois(wide), MASK, s)
    mixed = [narrow[0], wide[0], wide[1], narrow[1], wide[2], narrow[2]]
    G = check(hip_ctx, mixed, MASK, s)
    assert not diff(alone_n, G[[0, 3, 5]]), diff(alone_n, G[[0, 3, 5]])[:10]
    assert not diff(alone_w, G[[1, 2, 4]]), diff(alone_w, G[[1, 2, 4]])[:10]
