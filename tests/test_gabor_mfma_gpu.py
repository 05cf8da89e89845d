"""The MFMA screening stage of the Gabor kernel (roi_shape.hip, run_bands_mfma; round 5): the band-pass filters run four at a time
as v_mfma_f32_16x16x32_f16 over two planes of f16 digits, pixels whose screened energy lies inside the error band are recomputed with
the reference's arithmetic.  Count ratios must be the oracle's bit for bit -- on either side of every boundary the stage has:
the digit split (2^11, 2^16), the groups of four filters, the lists of band pixels (and their overflow), the one-wave and four-wave
kernels, the separable low-pass pass and its fallback."""
import numpy as np
import pytest

from nyxus_amd import _abi
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _same(G, O):
    return ((G == O) | (np.isnan(G) & np.isnan(O))).all()


def _bank(s, n, seed=0):
    rng = np.random.default_rng(seed)
    s.gabor_n_filters = n
    for i in range(n):
        s.gabor_f0[i] = float([4.0, 16.0, 32.0, 64.0][i % 4] * rng.uniform(0.9, 1.1)) if seed else [4.0, 16.0, 32.0, 64.0][i % 4]
        s.gabor_theta[i] = float(np.pi * i / n)
    return s


def _roi(w, h, v, ellipse=False):
    yy, xx = np.mgrid[0:h, 0:w]
    m = ((xx - w / 2 + .5) ** 2 / (w / 2) ** 2 + (yy - h / 2 + .5) ** 2 / (h / 2) ** 2 <= 1.0) if ellipse else np.ones((h, w), bool)
    if not m.any():
        m[0, 0] = True
    return dict(x=xx[m].astype(np.uint16), y=yy[m].astype(np.uint16), inten=np.broadcast_to(np.asarray(v), (h, w))[m].astype(np.uint32))


@pytest.mark.parametrize("top", [200, 2047, 2048, 4095, 40000, 65535, 65536, 1 << 20])
def test_digit_split_boundaries(hip_ctx, top):
    """Intensities below 2^11 take one digit plane, below 2^16 two (main: the top eleven significant bits), from 2^16 on the
    packed-fp32 pass -- every depth must give the reference's counts, in the four-wave kernel (a 61-pixel disk in the batch) and in the
    one-wave kernel of small ROIs."""
    rng = np.random.default_rng(top)
    big = [_roi(61, 61, rng.integers(1, top + 1, (61, 61)), True), _roi(40, 33, rng.integers(top // 2, top + 1, (33, 40))),
           _roi(17, 50, (top * (0.5 + 0.4 * np.sin(np.arange(17)[None, :] / 3.0) * np.cos(np.arange(50)[:, None] / 5.0))).astype(np.int64).clip(1))]
    big[0]["inten"][0] = top
    small = [_roi(w, h, rng.integers(1, top + 1, (h, w)), k % 2 == 0) for k, (w, h) in enumerate([(10, 9), (16, 14), (3, 12), (13, 1), (1, 1), (12, 16)])]
    small[1]["inten"][0] = top
    for rois in (big + small, small):
        b = _abi.batch_from_rois(rois)
        for s in (_abi.default_settings(8), _bank(_abi.default_settings(8), 8)):
            assert _same(hip_ctx.featurize_host(b, _abi.FAM_GABOR, s), po.oracle_featurize(b, _abi.FAM_GABOR, s))


def test_saturated_sixteen_bit_windows(hip_ctx):
    """The fp32 sliding box low-pass (default bank, intensities < 2^16): a 16 x 16 window of 65534 / 65535 sums to within 256 of 2^24,
    so the slide must subtract the leaving column before it adds the entering one (add-first passes 2^24, where fp32 drops odd
    integers: the low-pass maximum moved by one and every filter's threshold with it)."""
    rng = np.random.default_rng(65535)
    yy, xx = np.mgrid[0:48, 0:60]
    v1 = rng.integers(65534, 65536, (48, 60))                                  # every window near saturation, odd sums everywhere
    v2 = np.where((xx >= 10) & (xx < 45) & (yy >= 8) & (yy < 40), 65535 - (xx + yy) % 2, rng.integers(1, 3000, (48, 60)))
    v3 = np.full((40, 40), 65535); v3[::7, ::5] = 65534
    v4 = rng.integers(61000, 65536, (33, 47))
    rois = [_roi(60, 48, v1), _roi(60, 48, v2), _roi(40, 40, v3), _roi(47, 33, v4), _roi(60, 48, v1, True), _roi(15, 13, v1[:13, :15])]
    b = _abi.batch_from_rois(rois)
    for s in (_abi.default_settings(8), _bank(_abi.default_settings(8), 8)):
        assert _same(hip_ctx.featurize_host(b, _abi.FAM_GABOR, s), po.oracle_featurize(b, _abi.FAM_GABOR, s))


@pytest.mark.parametrize("nf", [1, 3, 4, 5, 8, 9, 16])
def test_filter_groups(hip_ctx, nf):
    """Filters run in groups of four operand columns: full groups, a last group of one to three, one filter alone."""
    rng = np.random.default_rng(nf)
    rois = [_roi(37, 29, rng.integers(1, 4096, (29, 37)), True), _roi(20, 20, rng.integers(0, 1 << 16, (20, 20))), _roi(9, 11, rng.integers(1, 256, (11, 9)))]
    b = _abi.batch_from_rois(rois)
    s = _bank(_abi.default_settings(8), nf, seed=nf)
    assert _same(hip_ctx.featurize_host(b, _abi.FAM_GABOR, s), po.oracle_featurize(b, _abi.FAM_GABOR, s))


@pytest.mark.parametrize("thr", [0.0, 0.025, 0.4, 1.0])
def test_band_lists_and_their_overflow(hip_ctx, thr):
    """Flat blocks put thousands of pixels at one energy: with that energy at the threshold every one of them lands in the band and
    the per-filter lists overflow (the filter is then recomputed over the whole box); a threshold of zero puts every zero response there."""
    yy, xx = np.mgrid[0:48, 0:56]
    rois = [_roi(56, 48, np.where((xx // 8 + yy // 8) % 2 == 0, 500, 3000)), _roi(56, 48, np.full((48, 56), 1234)),
            _roi(56, 48, np.where(xx < 28, 100, 60000)), _roi(30, 30, np.where((xx[:30, :30] + yy[:30, :30]) % 2 == 0, 10, 4000), True)]
    v = np.full((48, 56), 777); v[20, 30] = 778
    rois.append(_roi(56, 48, v))
    b = _abi.batch_from_rois(rois)
    for nf in (4, 8):
        s = _bank(_abi.default_settings(8), nf) if nf == 8 else _abi.default_settings(8)
        s.gabor_graythr = thr
        assert _same(hip_ctx.featurize_host(b, _abi.FAM_GABOR, s), po.oracle_featurize(b, _abi.FAM_GABOR, s))


def test_box_shapes_around_the_tiles(hip_ctx):
    """Sixteen box rows make a row tile, four columns a column group (the first one starts at column -1): widths and heights on
    either side of those, rows and columns alone."""
    rng = np.random.default_rng(5)
    rois = []
    for w in (1, 2, 3, 4, 5, 7, 8, 9, 31, 32, 33, 63, 64, 65):
        for h in (1, 2, 15, 16, 17, 31, 32, 33, 47):
            if (w * h) % 3 == 0 or w < 6 or h < 3:
                rois.append(_roi(w, h, rng.integers(1, 4096, (h, w)), (w + h) % 5 == 0))
    b = _abi.batch_from_rois(rois)
    s = _abi.default_settings(8)
    assert _same(hip_ctx.featurize_host(b, _abi.FAM_GABOR, s), po.oracle_featurize(b, _abi.FAM_GABOR, s))


def test_low_pass_that_does_not_factor(hip_ctx):
    """The separable low-pass pass needs tap(j, i) = C_j B_i (ensure_gabor_bank checks the taps it built).  gamma = 1 and the default
    gamma factor alike; a bank whose low-pass wave length makes the check fail would fall back to the full pass -- same counts either way
    (NYXHIP_GABOR_NO_LPSEP=1 forces the fallback in tools/gabor_fuzz.py runs)."""
    rng = np.random.default_rng(6)
    rois = [_roi(33, 27, rng.integers(1, 4096, (27, 33)), True), _roi(12, 12, rng.integers(1, 60000, (12, 12)))]
    b = _abi.batch_from_rois(rois)
    for gamma, f0lp, s2l in ((0.1, 0.1, 0.8), (1.0, 0.1, 0.8), (0.5, 2.0, 0.4), (0.1, 30.0, 1.5)):
        s = _abi.default_settings(8)
        s.gabor_gamma, s.gabor_f0lp, s.gabor_sig2lam = gamma, f0lp, s2l
        assert _same(hip_ctx.featurize_host(b, _abi.FAM_GABOR, s), po.oracle_featurize(b, _abi.FAM_GABOR, s))


def test_largest_lds_resident_boxes(hip_ctx):
    """The in-place rewrite of the fp32 plane as digit planes holds the plane in registers: 32 words per thread up to 8192 words, 64
    beyond (boxes of ~75 x 75 up to the LDS bound) -- both builds, next to a box of the first kind."""
    rng = np.random.default_rng(8)
    rois = [_roi(100, 90, rng.integers(1, 4096, (90, 100)), True), _roi(126, 119, rng.integers(1, 60000, (119, 126))),
            _roi(74, 76, rng.integers(1, 256, (76, 74))), _roi(61, 61, rng.integers(1, 4096, (61, 61)), True)]
    b = _abi.batch_from_rois(rois)
    for s in (_abi.default_settings(8), _bank(_abi.default_settings(8), 8)):
        assert _same(hip_ctx.featurize_host(b, _abi.FAM_GABOR, s), po.oracle_featurize(b, _abi.FAM_GABOR, s))


def test_mfma_rounding_model_holds_on_this_device(tmp_path):
    """The error band of the MFMA screening stage (roi_shape.hip: kErr = 1.85e-5 a_max) rests on how v_mfma_f32_16x16x32_f16 adds up --
    products summed before ONE round-to-nearest-even, terms cut to a grid of 2^-24 of the largest -- and takes 33 x 2^-24 of
    (|C| + sum |a b|) per instruction.  tools/mfma_rounding_probe.hip measures that behaviour (round 5: worst 3.1 x 2^-24); another
    firmware / ROCm that adds differently must fail HERE, not flip a Gabor count somewhere."""
    import os
    import re
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "mfma_rounding_probe")
    subprocess.run([hipcc, "-O2", "--offload-arch=gfx950", "-o", exe, os.path.join(root, "tools", "mfma_rounding_probe.hip")], check=True, timeout=600)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=300).stdout
    m1 = re.search(r"1\. .*D - 2\^24 = (\d+)", out)
    assert m1 and int(m1.group(1)) == 32, out                      # the 32 products are summed before the rounding
    m2 = re.search(r"2\. .*D - 2\^24 = ([-\d ]+?)  \(", out)
    assert m2 and m2.group(1).split() == ["0", "4", "4", "8", "-1", "-3"], out   # round to nearest even
    m3 = re.search(r"3\. .*exact up to s = (\d+)", out)
    assert m3 and int(m3.group(1)) >= 12, out                      # >= 24 bits kept below the largest term
    m4 = re.search(r"4\. .* = ([0-9.eE+-]+) of \(\|C\| \+ sum \|a b\|\)", out)
    assert m4, out
    worst = float(m4.group(1))
    assert worst <= 33.0 * 2.0 ** -24, f"observed {worst:.3g} of (|C| + sum |a b|) per instruction; the stage's band assumes 33 x 2^-24 = {33.0 * 2.0 ** -24:.3g}\n{out}"
