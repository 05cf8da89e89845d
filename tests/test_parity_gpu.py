"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle
(and, when the prebuilt checker travelled, the reference's own classes).

Bar (north_star / BASELINE.md 3.6): bit-exact for label/count/order-statistic features,
<= 1e-5 relative elsewhere (tests/parity.py)."""
import numpy as np
import pytest

from nyxus_amd import _abi, _lib
from oracle import pyoracle as po
from tests import fixtures, parity, synth

pytestmark = pytest.mark.gpu

MASK = _abi.FAM_INTENSITY | _abi.FAM_GLCM
CONFIGS = [(8, False, 4), (64, False, 4), (-16, False, 4), (100, False, 2), (20, True, 4), (24, False, 1)]


def _check(ctx, b, mask, s, against_ref=True):
    names = _lib.column_names(mask, s)
    G = ctx.featurize_host(b, mask, s)
    O = po.oracle_featurize(b, mask, s)
    bad = parity.compare_tables(G, O, names, batch=b)
    assert not bad, "\n".join(bad[:20])
    if against_ref and po.have_ref():
        R = po.ref_featurize(b, mask, s, n_threads=2)
        bad = parity.compare_tables(G, R, names, batch=b)
        assert not bad, "vs reference classes:\n" + "\n".join(bad[:20])
    return G


@pytest.mark.parametrize("gd,ibsi,na", CONFIGS)
def test_random_rois_match_oracle(hip_ctx, gd, ibsi, na):
    rois = synth.random_rois(120, seed=11, slide=True)
    if ibsi:
        rois = [dict(r, inten=(np.asarray(r["inten"]) % 9).astype(np.uint32)) for r in rois]
    s = _abi.default_settings(gd, ibsi)
    s.glcm_n_angles = na
    _check(hip_ctx, _abi.batch_from_rois(rois), MASK, s)


@pytest.mark.parametrize("mask", [_abi.FAM_INTENSITY, _abi.FAM_GLCM])
def test_single_family_masks(hip_ctx, mask):
    s = _abi.default_settings(8)
    _check(hip_ctx, _abi.batch_from_rois(synth.random_rois(40, seed=5)), mask, s)


@pytest.mark.parametrize("irregular", [False, True])
def test_benchmark_tile(hip_ctx, irregular):
    """One tile of the metric configuration (config 2+3): 196 ROIs of a 1024^2 tile."""
    b = synth.tile_batch(1, irregular=irregular)
    s = _abi.default_settings(8)
    G = _check(hip_ctx, b, MASK, s)
    assert G.shape == (196, 185)


def test_edge_cases(hip_ctx):
    """Single pixel, two pixels, constant, all-zero, zero pixels inside, 0xFFFFFFFF."""
    rois = [
        dict(x=[5], y=[7], inten=[9]),
        dict(x=[0, 1], y=[0, 0], inten=[3, 3]),
        dict(x=[0, 1, 2, 3], y=[0, 0, 0, 0], inten=[0, 0, 0, 0]),
        dict(x=[0, 1, 2, 0, 1, 2], y=[0, 0, 0, 1, 1, 1], inten=[0, 5, 0, 7, 0, 9]),
        dict(x=[0, 1, 0, 1], y=[0, 0, 1, 1], inten=[2 ** 32 - 1, 2 ** 32 - 2, 1, 2 ** 31]),
        dict(x=list(range(64)), y=[0] * 64, inten=[42] * 64),
        dict(x=[0, 3], y=[0, 3], inten=[1, 200]),          # bbox mostly background
    ]
    for gd in (8, -8):
        s = _abi.default_settings(gd)
        _check(hip_ctx, _abi.batch_from_rois(rois), MASK, s)


def test_reference_golden_fixtures_through_hip(hip_ctx):
    """The reference's own fixtures (tests/golden/reference_tests.json) through the HIP
    path at the reference's own tolerances (rel 1e-3 / 1e-2)."""
    ref = fixtures.reference_tests()
    r = fixtures.roi_from_triplets(ref["pixels"]["pixelIntensityFeaturesTestData"])
    r["slide_min"], r["slide_max"] = 0.0, 65535.0
    b = _abi.batch_from_rois([r])
    s = _abi.default_settings(24)
    names = _lib.column_names(_abi.FAM_INTENSITY, s)
    row = dict(zip(names, hip_ctx.featurize_host(b, _abi.FAM_INTENSITY, s)[0]))
    for k, v in ref["goldens"]["firstorder_2d_matlab_ref_vals"].items():
        if k != "UNIFORMITY":
            assert abs(row[k] - v) <= abs(v) / 1000.0, (k, row[k], v)
    # GLCM regression (matlab binning, 100 levels) and IBSI consensus values
    for gold, s in (("glcm_2d_regression_ref_vals", _abi.default_settings(100)),
                    ("glcm_2d_ibsi_ref_vals", _abi.default_settings(0, ibsi=True))):
        s.glcm_grey_depth = s.grey_depth
        s.grey_depth = s.grey_depth or 1  # histogram bins unused for GLCM-only
        T = hip_ctx.featurize_host(fixtures.ibsi_phantom_batch(ref), _abi.FAM_GLCM, s)
        gn = _lib.column_names(_abi.FAM_GLCM, s)
        for k, v in ref["goldens"][gold].items():
            cols = [i for i, n in enumerate(gn) if n in (f"{k}_0", f"{k}_45", f"{k}_90", f"{k}_135")]
            got = T[:, cols].sum() / 16.0
            assert abs(got - v) <= abs(v) / 100.0, (gold, k, got, v)


TEX = _abi.FAM_GLRLM | _abi.FAM_GLSZM | _abi.FAM_NGTDM


@pytest.mark.parametrize("gd,ibsi", [(8, False), (64, False), (-16, False), (20, True), (100, False)])
def test_texture_families_random_rois(hip_ctx, gd, ibsi):
    """GLRLM + GLSZM + NGTDM (roi_texture.hip) vs oracle and the reference classes."""
    rois = synth.random_rois(100, seed=21)
    if ibsi:
        rois = [dict(r, inten=(np.asarray(r["inten"]) % 9).astype(np.uint32)) for r in rois]
        rois = [r for r in rois if np.asarray(r["inten"]).max() > 0]  # reference UB on all-zero IBSI NGTDM (ngtdm.cpp:58)
    _check(hip_ctx, _abi.batch_from_rois(rois), TEX, _abi.default_settings(gd, ibsi))


@pytest.mark.parametrize("fam", [_abi.FAM_GLRLM, _abi.FAM_GLSZM, _abi.FAM_NGTDM])
def test_texture_single_family(hip_ctx, fam):
    _check(hip_ctx, _abi.batch_from_rois(synth.random_rois(40, seed=2)), fam, _abi.default_settings(8))


def test_all_five_families_on_benchmark_tile(hip_ctx):
    """Config 4 feature set (*ALL_GLCM*+*ALL_GLRLM*+*ALL_GLSZM*+*ALL_NGTDM*+*ALL_INTENSITY*) on one tile."""
    b = synth.tile_batch(3, irregular=True)
    G = _check(hip_ctx, b, MASK | TEX, _abi.default_settings(8))
    assert G.shape == (196, 185 + 80 + 16 + 5)


def test_texture_goldens_through_hip(hip_ctx):
    """Reference golden tables (tests/test_2d_{glrlm,glszm,ngtdm}_{regression,ibsi}.h) through the HIP path."""
    ref = fixtures.reference_tests()
    b = fixtures.ibsi_phantom_batch(ref)
    cases = [(_abi.FAM_GLRLM, "glrlm_2d_regression_ref_vals", 100, False, True), (_abi.FAM_GLRLM, "glrlm_2d_ibsi_ref_vals", 128, True, True),
             (_abi.FAM_GLSZM, "glszm_2d_regression_ref_vals", 64, False, False), (_abi.FAM_GLSZM, "glszm_2d_ibsi_ref_vals", 128, True, False),
             (_abi.FAM_NGTDM, "ngtdm_2d_regression_ref_vals", 100, False, False), (_abi.FAM_NGTDM, "ngtdm_2d_ibsi_ref_vals", 128, True, False)]
    for fam, gold, gd, ibsi, angled in cases:
        s = _abi.default_settings(gd, ibsi)
        T = hip_ctx.featurize_host(b, fam, s)
        names = _lib.column_names(fam, s)
        for k, v in ref["goldens"][gold].items():
            cols = [i for i, n in enumerate(names) if n == k or n in (f"{k}_0", f"{k}_45", f"{k}_90", f"{k}_135")]
            got = T[:, cols].sum() / (16.0 if angled else 4.0)
            assert abs(got - v) <= abs(v) / 100.0, (gold, k, got, v)


def test_device_resident_async_path(hip_ctx):
    """Device pointers + async launch on torch's current stream (the bench path)."""
    import torch
    b = synth.tile_batch(2)
    s = _abi.default_settings(8)
    ncol = hip_ctx.n_columns(MASK, s)
    dev = {k: torch.from_numpy(getattr(b, k).view(np.int16 if getattr(b, k).dtype == np.uint16 else
                                                  np.int32 if getattr(b, k).dtype == np.uint32 else np.int64)).cuda()
           for k in ("px_offset", "x", "y", "inten", "bbox_w", "bbox_h", "min_inten", "max_inten")}
    out = torch.empty((b.n_roi, ncol), dtype=torch.float64, device="cuda")
    cb = _abi.Batch()
    cb.n_roi = b.n_roi
    for k, t in dev.items():
        setattr(cb, k, t.data_ptr())
    cb.memory = _abi.MEM_DEVICE
    hip_ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        hip_ctx.featurize_device_async(cb, MASK, s, out.data_ptr(), ncol)  # extrema derived on device
        hip_ctx.sync()
    finally:
        hip_ctx._lib.nyxhip_set_stream(hip_ctx._h, None)
        hip_ctx._lib.nyxhip_set_stream  # keep user-stream mode: NULL = default stream
    G = out.cpu().numpy()
    O = po.oracle_featurize(b, MASK, s)
    assert not parity.compare_tables(G, O, _lib.column_names(MASK, s))


def _large_rois(seed=5):
    rng = np.random.default_rng(seed)
    rois = synth.random_rois(12, seed=seed)
    n = 70000  # > LDS capacity of the sort buffer (2^17 * 4 B): served from the global workspace
    rois.insert(3, dict(x=np.arange(n) % 300, y=np.arange(n) // 300, inten=rng.integers(1, 1000, n)))
    yy, xx = np.mgrid[0:420, 0:380]
    keep = ((xx - 190.0) ** 2 / 190.0**2 + (yy - 210.0) ** 2 / 210.0**2) <= 1.0
    keep &= rng.random(keep.shape) > 0.02  # pinholes -> zones and runs are interrupted
    rois.append(dict(x=xx[keep], y=yy[keep], inten=rng.integers(0, 60000, int(keep.sum()))))
    return rois


@pytest.mark.parametrize("gd", [8, -20])
def test_rois_beyond_lds_use_global_workspace(hip_ctx, gd):
    """ROIs larger than the 160 KiB carve-out are re-run with their scratch in HBM; the small ROIs of the
    same batch stay on the LDS path.  Same parity bar as everything else."""
    rois = _large_rois()
    s = _abi.default_settings(gd)
    mask = _abi.FAM_ALL & ~_abi.FAM_GABOR
    if gd < 0:      # GLDZM / NGLDM refuse radiomics binning (undefined in the reference)
        mask &= ~(_abi.FAM_GLDZM | _abi.FAM_NGLDM)
    b = _abi.batch_from_rois(rois)
    G = hip_ctx.featurize_host(b, mask, s)
    O = po.oracle_featurize(b, mask, s)
    assert not parity.compare_tables(G, O, _lib.column_names(mask, s), batch=b)


def test_large_roi_gabor_is_exact(hip_ctx):
    rois = _large_rois(seed=9)[2:5]
    s = _abi.default_settings(8)
    b = _abi.batch_from_rois(rois)
    G = hip_ctx.featurize_host(b, _abi.FAM_GABOR, s)
    O = po.oracle_featurize(b, _abi.FAM_GABOR, s)
    assert np.array_equal(G, O)


def test_bad_family_mask_is_an_error(hip_ctx):
    b = _abi.batch_from_rois(synth.random_rois(3))
    with pytest.raises(_lib.NyxHipError) as ei:
        hip_ctx.featurize_host(b, 1 << 12, _abi.default_settings(8))
    assert ei.value.code == 1


SHAPE = _abi.FAM_GABOR | _abi.FAM_ZERNIKE


def _bank8(s):
    """8-filter bank of BASELINE.json configs[4]: theta in {0, 22.5, ..., 157.5} deg, f0 cycling {4,16,32,64}."""
    s.gabor_n_filters = 8
    for i in range(8):
        s.gabor_f0[i] = [4.0, 16.0, 32.0, 64.0][i % 4]
        s.gabor_theta[i] = np.pi * i / 8
    return s


def test_gabor_zernike_reference_goldens_through_hip(hip_ctx):
    """tests/test_gabor_truth.h:27-47 (4 DSB2018 ROIs x default bank, exact counts) and
    tests/test_2d_zernike_regression.h:12-24 (abs 1e-9) through the HIP path."""
    ref = fixtures.reference_tests()
    b = _abi.batch_from_rois([fixtures.dsb_roi(d) for d in ref["dsb2018"]])
    G = hip_ctx.featurize_host(b, _abi.FAM_GABOR, _abi.default_settings(64))
    assert np.array_equal(G, np.array(ref["gabor_truth"]))      # count ratios: bit-exact
    r = fixtures.roi_from_triplets(ref["pixels"]["shape2d_morphology_intensity"], ref["pixels"]["shape2d_morphology_mask"])
    z = hip_ctx.featurize_host(_abi.batch_from_rois([r]), _abi.FAM_ZERNIKE, _abi.default_settings(128))[0]
    want = np.array(ref["vector_goldens"]["zernike_2d_regression_ref_vals"]["ZERNIKE2D"])
    assert np.all(np.abs(z - want) <= 1e-9)


def test_gabor_zernike_random_and_dsb_rois(hip_ctx):
    """Config 5 shape: DSB2018-sized ROIs, 8-orientation bank + ZERNIKE2D; Gabor is compared bit-exact."""
    ref = fixtures.reference_tests()
    rois = synth.random_rois(60, seed=4, rmax=12) + [fixtures.dsb_roi(d) for d in ref["dsb2018"]]
    b = _abi.batch_from_rois(rois)
    s = _bank8(_abi.default_settings(64))
    names = _lib.column_names(SHAPE, s)
    G = hip_ctx.featurize_host(b, SHAPE, s)
    O = po.oracle_featurize(b, SHAPE, s)
    gab = [i for i, n in enumerate(names) if n.startswith("GABOR")]
    assert np.array_equal(G[:, gab], O[:, gab]) or ((G[:, gab] == O[:, gab]) | (np.isnan(G[:, gab]) & np.isnan(O[:, gab]))).all()
    assert not parity.compare_tables(G, O, names)
    if po.have_ref():
        assert not parity.compare_tables(G, po.ref_featurize(b, SHAPE, s, 2), names)


def test_all_seven_families_one_call(hip_ctx):
    b = synth.tile_batch(4, irregular=True, size=512)
    s = _bank8(_abi.default_settings(8))
    G = _check(hip_ctx, b, _abi.FAM_NORTH_STAR, s, against_ref=False)
    assert G.shape[1] == 185 + 80 + 16 + 5 + 8 + 30


# ---- GLDZM + GLDM + NGLDM (roi_dependence.hip) ------------------------------------------------------
DEP = _abi.FAM_GLDZM | _abi.FAM_GLDM | _abi.FAM_NGLDM


@pytest.mark.parametrize("gd,ibsi", [(8, False), (64, False), (20, True), (100, False)])
def test_dependence_families_random_rois(hip_ctx, gd, ibsi):
    rois = synth.random_rois(100, seed=23)
    if ibsi:
        rois = [dict(r, inten=(np.asarray(r["inten"]) % 9).astype(np.uint32)) for r in rois]
    _check(hip_ctx, _abi.batch_from_rois(rois), DEP, _abi.default_settings(gd, ibsi))


def test_dependence_radiomics_binning(hip_ctx):
    """Negative grey depth: GLDM against oracle + reference classes; GLDZM (the reference indexes one row past its matrix
    for level-0 zones, gldzm.cpp:44-50) and NGLDM (GREYDEPTH cast to unsigned, ngldm.cpp:201) are refused."""
    b = _abi.batch_from_rois(synth.random_rois(60, seed=4))
    s = _abi.default_settings(-16)
    _check(hip_ctx, b, _abi.FAM_GLDM, s)
    for fam in (_abi.FAM_NGLDM, _abi.FAM_GLDZM):
        with pytest.raises(_lib.NyxHipError) as ei:
            hip_ctx.featurize_host(b, fam, s)
        assert ei.value.code == 1


@pytest.mark.parametrize("fam", [_abi.FAM_GLDZM, _abi.FAM_GLDM, _abi.FAM_NGLDM])
def test_dependence_single_family_and_column_interleave(hip_ctx, fam):
    """Each family alone, and together with the texture kernel's families whose columns it sits between
    (Feature2D order: GLRLM, GLDZM, GLSZM, GLDM, NGLDM, NGTDM)."""
    b = _abi.batch_from_rois(synth.random_rois(40, seed=2))
    s = _abi.default_settings(8)
    _check(hip_ctx, b, fam, s)
    G = _check(hip_ctx, b, fam | TEX | _abi.FAM_INTENSITY, s)
    names = _lib.column_names(fam | TEX | _abi.FAM_INTENSITY, s)
    order = [n.split("_")[0] for n in names if n.split("_")[0] in ("GLRLM", "GLDZM", "GLSZM", "GLDM", "NGLDM", "NGTDM")]
    dedup = [k for i, k in enumerate(order) if i == 0 or order[i - 1] != k]
    assert dedup == [k for k in ("GLRLM", "GLDZM", "GLSZM", "GLDM", "NGLDM", "NGTDM") if k in dedup]
    assert G.shape[1] == len(names)


def test_dependence_goldens_through_hip(hip_ctx):
    """tests/test_2d_gldm_{ibsi,regression}.h, test_2d_ngldm_{ibsi,mirp,regression}.h, test_2d_gldzm_ibsi.h through HIP."""
    ref = fixtures.reference_tests()
    b = fixtures.ibsi_phantom_batch(ref)
    s = _abi.default_settings(128, True)
    for fam, gold, tol in [(_abi.FAM_GLDM, "gldm_2d_ibsi_ref_vals", 100.0), (_abi.FAM_NGLDM, "ngldm_2d_ibsi_ref_vals", 100.0),
                           (_abi.FAM_NGLDM, "ngldm_2d_mirp_ref_vals", 1e9), (_abi.FAM_NGLDM, "ngldm_2d_regression_ref_vals", 1e9),
                           (_abi.FAM_GLDZM, "gldzm_2d_ibsi_ref_vals", 2.0)]:
        T = hip_ctx.featurize_host(b, fam, s)
        names = _lib.column_names(fam, s)
        for k, v in ref["goldens"][gold].items():
            got = T[:, names.index(k)].sum() / 4.0
            assert abs(got - v) <= abs(v / tol), (gold, k, got, v)
    cat = _abi.batch_from_rois([fixtures.roi_from_triplets(ref["pixels"]["cat2500_int"], ref["pixels"]["cat2500_seg"])])
    s = _abi.default_settings(128, False)
    T = hip_ctx.featurize_host(cat, _abi.FAM_GLDM, s)[0]
    names = _lib.column_names(_abi.FAM_GLDM, s)
    for k, v in ref["goldens"]["gldm_2d_regression_ref_vals"].items():
        assert abs(T[names.index(k)] - v) <= abs(v / 1000.0), (k, T[names.index(k)], v)


def test_dependence_large_rois_and_serpentine_zone(hip_ctx):
    """ROIs beyond the LDS carve-out (global workspace) and a spiral-shaped zone (worst case for label sweeps)."""
    rois = _large_rois(seed=7)
    n = 41
    sp = np.zeros((n, n), np.uint32)       # square spiral: one long 4-connected corridor of level A in a field of level B
    dy, dx = 0, 1
    yy = xx = 0
    for _ in range(n * n):
        sp[yy, xx] = 1
        ny, nx = yy + dy, xx + dx
        ahead2y, ahead2x = yy + 2 * dy, xx + 2 * dx
        if not (0 <= ny < n and 0 <= nx < n) or (0 <= ahead2y < n and 0 <= ahead2x < n and sp[ahead2y, ahead2x]):
            dy, dx = dx, -dy
            ny, nx = yy + dy, xx + dx
            if not (0 <= ny < n and 0 <= nx < n) or sp[ny, nx] or (0 <= yy + 2 * dy < n and 0 <= xx + 2 * dx < n and sp[yy + 2 * dy, xx + 2 * dx]):
                break
        yy, xx = ny, nx
    gy, gx = np.mgrid[0:n, 0:n]
    rois.append(dict(x=gx.ravel(), y=gy.ravel(), inten=np.where(sp.ravel() > 0, 4000, 100).astype(np.uint32)))
    b = _abi.batch_from_rois(rois)
    _check(hip_ctx, b, DEP, _abi.default_settings(8), against_ref=False)


def test_all_twelve_families_one_call(hip_ctx):
    b = synth.tile_batch(4, irregular=True, size=512)
    s = _bank8(_abi.default_settings(8))
    G = _check(hip_ctx, b, _abi.FAM_ALL, s, against_ref=False)
    assert G.shape[1] == 185 + 80 + 18 + 16 + 14 + 19 + 5 + 8 + 30 + 90 + 90


# ---- SURVEY 8(f) #4, last item: contour + 2-D geometric moments (roi_moments.hip) ---------------------------------------
MOM = _abi.FAM_SMOMS | _abi.FAM_IMOMS


_moment_atol = parity.moment_atol


def _check_moments(ctx, b, mask=MOM, against_ref=True):
    s = _abi.default_settings(8)
    names = _lib.column_names(mask, s)
    G = ctx.featurize_host(b, mask, s)
    O = po.oracle_featurize(b, mask, s)
    bad = parity.compare_tables(G, O, names, batch=b)
    assert not bad, "\n".join(bad[:20])
    if against_ref and po.have_ref():
        bad = parity.compare_tables(G, po.ref_featurize(b, mask, s, n_threads=2), names, batch=b)
        assert not bad, "vs reference classes:\n" + "\n".join(bad[:20])
    return G, names


@pytest.mark.parametrize("seed,rmax", [(9, 25), (3, 12), (5, 40)])
def test_geomoments_random_rois(hip_ctx, seed, rmax):
    """Irregular ROIs with holes, specks and single pixels: the contour walk (bifurcations, failed chains, X-crossings) and
    both moment families against the oracle and the reference classes."""
    _check_moments(hip_ctx, _abi.batch_from_rois(synth.random_rois(60, seed=seed, rmax=rmax)))


@pytest.mark.parametrize("fam", [_abi.FAM_SMOMS, _abi.FAM_IMOMS])
def test_geomoments_single_family_and_with_others(hip_ctx, fam):
    b = _abi.batch_from_rois(synth.random_rois(30, seed=14))
    _check_moments(hip_ctx, b, fam)
    _check_moments(hip_ctx, b, fam | _abi.FAM_INTENSITY | _abi.FAM_ZERNIKE | _abi.FAM_GLDM, against_ref=False)


def test_geomoments_reference_goldens_through_hip(hip_ctx):
    """tests/test_2d_moments_regression.h and test_2d_moments_skimage.h (rectangle + wedge fixtures, scikit-image goldens,
    |actual - golden| <= 1e-6 * max(1, |golden|, |actual|)) through the HIP path."""
    ref = fixtures.reference_tests()
    b = _abi.batch_from_rois([fixtures.geomoment_rectangle_roi(), fixtures.geomoment_wedge_roi()])
    s = _abi.default_settings(256)
    T = hip_ctx.featurize_host(b, MOM, s)
    names = _lib.column_names(MOM, s)
    assert names == fixtures.SMOM_NAMES + fixtures.IMOM_NAMES
    for lst, row in [("moments_2d_regression_shape_ref_vals", 0), ("moments_2d_regression_intensity_ref_vals", 0),
                     ("moments_2d_skimage_shape_ref_vals", 0), ("moments_2d_skimage_intensity_ref_vals", 0),
                     ("moments_2d_skimage_normraw_shape_ref_vals", 0), ("moments_2d_skimage_normraw_intensity_ref_vals", 0),
                     ("moments_2d_skimage_wedge_hu_ref_vals", 1)]:
        for k, v in ref["moment_goldens"][lst].items():
            got = T[row, names.index(k)]
            assert np.isfinite(got) and abs(got - v) <= 1e-6 * max(1.0, abs(v), abs(got)), (lst, k, got, v)


def test_geomoments_benchmark_tile_and_large_rois(hip_ctx):
    _check_moments(hip_ctx, synth.tile_batch(2, irregular=True), against_ref=False)
    _check_moments(hip_ctx, _abi.batch_from_rois(_large_rois(seed=3)), against_ref=False)   # contour planes beyond LDS


def test_geomoments_adversarial_masks(hip_ctx):
    """Contour logic on shapes that stress it: noise at several densities, checkerboards, crosses, rings, striped noise,
    thick diagonals with specks (tools/contour_fuzz.py runs the long version).  Rows whose weighted mass
    w00 = sum I log(d + eps) cancels to < 5 % of m00 are left out: their normalised weighted columns divide by powers of w00."""
    rng = np.random.default_rng(2)
    rois = []
    for k in range(300):
        h, w = rng.integers(1, 40, 2)
        kind = k % 6
        yy, xx = np.mgrid[0:h, 0:w]
        if kind == 0:
            m = rng.random((h, w)) < rng.choice([0.1, 0.3, 0.5, 0.7, 0.9])
        elif kind == 1:
            m = (xx + yy) % 2 == 0
        elif kind == 2:
            m = np.zeros((h, w), bool); m[rng.integers(0, h), :] = True; m[:, rng.integers(0, w)] = True
        elif kind == 3:
            r = np.hypot(xx - w / 2, yy - h / 2); m = (r < min(h, w) / 2) & (r > min(h, w) / 4)
        elif kind == 4:
            m = rng.random((h, w)) < 0.6; m[1:-1:2, :] = False
        else:
            m = (np.abs(xx - yy) <= 1) | (rng.random((h, w)) < 0.05)
        if not m.any():
            m[0, 0] = True
        ys, xs = np.nonzero(m)
        rois.append(dict(x=xs - xs.min(), y=ys - ys.min(), inten=rng.integers(1, 500, len(xs)).astype(np.uint32)))
    b = _abi.batch_from_rois(rois)
    s = _abi.default_settings(8)
    names = _lib.column_names(MOM, s)
    G = hip_ctx.featurize_host(b, MOM, s)
    O = po.oracle_featurize(b, MOM, s)
    ok = (np.abs(O[:, names.index("WEIGHTED_SPAT_MOMENT_00")]) >= 0.05 * O[:, names.index("SPAT_MOMENT_00")]) & \
         (np.abs(O[:, names.index("IMOM_WRM_00")]) >= 0.05 * O[:, names.index("IMOM_RM_00")])
    assert ok.sum() > 200
    atol = {k: v[ok] for k, v in parity.moment_atol(b, O, names).items()}
    bad = parity.compare_tables(G[ok], O[ok], names, atol=atol)
    assert not bad, "\n".join(bad[:20])


@pytest.mark.parametrize("gd", [255, 256])
def test_glcm_grey_depth_beyond_lds_runs_from_the_global_workspace(hip_ctx, gd):
    """A 256-level co-occurrence matrix (4 x 256 KiB for the four angles) cannot sit in a CU's LDS: the INTENSITY + GLCM
    group then runs with its scratch in HBM for every ROI; the other families of the call keep their LDS kernels."""
    b = _abi.batch_from_rois(synth.random_rois(24, seed=31))
    s = _abi.default_settings(gd)
    _check(hip_ctx, b, MASK, s)
    _check(hip_ctx, b, MASK | _abi.FAM_GLRLM | _abi.FAM_NGTDM, s, against_ref=False)


@pytest.mark.parametrize("vmax", [200, 255, 1000])
def test_ibsi_glcm_of_order_beyond_128(hip_ctx, vmax):
    """ibsi=True on an 8-bit image (and a 10-bit one): the co-occurrence matrix has the order of the largest intensity
    (/root/reference/src/nyx/features/glcm.cpp:400-419 allocates max x max).  Orders that do not fit LDS run from the global
    workspace, like a matlab grey depth of 256 does."""
    rng = np.random.default_rng(vmax)
    rois = []
    for k in range(20):
        r = synth.random_rois(1, seed=100 + k, rmax=18)[0]
        v = rng.integers(0 if k % 4 == 0 else 1, vmax + 1, len(r["inten"])).astype(np.uint32)
        if k == 3:
            v[0] = vmax
        rois.append(dict(r, inten=v))
    s = _abi.default_settings(64, True)
    b = _abi.batch_from_rois(rois)
    _check(hip_ctx, b, MASK, s)
    _check(hip_ctx, b, _abi.FAM_GLCM, s, against_ref=False)


def test_ibsi_texture_levels_beyond_255(hip_ctx):
    """ibsi=True on 10-bit intensities: the texture and dependence families take the intensities themselves as levels (no binning);
    the group's largest intensity comes with the class header and sizes the level tables (LDS when they fit, else the workspace)."""
    rng = np.random.default_rng(12)
    rois = []
    for k in range(16):
        r = synth.random_rois(1, seed=300 + k, rmax=10)[0]
        rois.append(dict(r, inten=rng.integers(1, 1001 if k % 2 else 301, len(r["inten"])).astype(np.uint32)))
    s = _abi.default_settings(64, True)
    b = _abi.batch_from_rois(rois)
    mask = _abi.FAM_GLRLM | _abi.FAM_GLSZM | _abi.FAM_NGTDM | _abi.FAM_GLDM | _abi.FAM_NGLDM | _abi.FAM_GLDZM
    _check(hip_ctx, b, mask, s, against_ref=False)


def _edge_shape_rois(seed=31):
    rng = np.random.default_rng(seed)
    rois = []
    for (w, h) in [(1, 1), (1, 40), (40, 1), (2, 3), (7, 9), (8, 8), (9, 17), (15, 16), (16, 15), (17, 64), (64, 64), (65, 33), (33, 70)]:
        m = np.ones((h, w), bool)
        if w > 4 and h > 4:
            m &= rng.random((h, w)) > 0.2
            m[0, 0] = m[h - 1, w - 1] = True          # keep the bounding box
        ys, xs = np.nonzero(m)
        rois.append(dict(x=xs, y=ys, inten=rng.integers(1, 60000, len(xs)).astype(np.uint32)))
    yy, xx = np.mgrid[0:61, 0:61]
    ys, xs = np.nonzero((xx - 30) ** 2 + (yy - 30) ** 2 <= 900)
    rois.append(dict(x=xs, y=ys, inten=rng.integers(1, 4096, len(xs)).astype(np.uint32)))   # the metric ROI shape
    return rois


def test_gabor_tiled_kernel_edge_shapes_exact(hip_ctx):
    """roi_gabor_tiled_kernel (16 x 16 bank): widths that are not a multiple of the 8-pixel tile, single rows / columns,
    boxes narrower than the kernel, the metric disk -- count ratios bit-exact against the oracle and the reference."""
    b = _abi.batch_from_rois(_edge_shape_rois())
    for s in (_abi.default_settings(8), _bank8(_abi.default_settings(8))):
        G = hip_ctx.featurize_host(b, _abi.FAM_GABOR, s)
        O = po.oracle_featurize(b, _abi.FAM_GABOR, s)
        assert ((G == O) | (np.isnan(G) & np.isnan(O))).all()
        if po.have_ref():
            Rf = po.ref_featurize(b, _abi.FAM_GABOR, s, 2)
            assert ((G == Rf) | (np.isnan(G) & np.isnan(Rf))).all()


def test_gabor_tile_dealing_boundaries_exact(hip_ctx):
    """Round 4 deals the tiles of the 256-thread Gabor kernel column-major in blocks of 16 rows (conflict-free 16-byte window reads),
    except where padding the box to a multiple of 16 rows would add a trip to the tile loop: heights on either side of 16 / 32 / 48 /
    64, widths on either side of the 8-pixel tile and of 64 / 128, smooth, noisy and flat fields -- count ratios bit-exact against the
    oracle for the default bank and the 8-orientation bank."""
    rng = np.random.default_rng(77)
    rois = []
    for k, (w, h) in enumerate([(61, 15), (61, 16), (61, 17), (40, 31), (40, 32), (40, 33), (64, 48), (65, 49), (57, 64), (56, 65), (120, 20), (129, 35),
                                (23, 100), (9, 129), (140, 61), (33, 47), (8, 16), (72, 33), (128, 17), (16, 64)]):
        yy, xx = np.mgrid[0:h, 0:w]
        m = ((xx - w / 2 + .5) ** 2 / (w / 2) ** 2 + (yy - h / 2 + .5) ** 2 / (h / 2) ** 2 <= 1.0) if k % 3 else np.ones((h, w), bool)
        if k % 4 == 0:
            v = rng.integers(1, 4096, (h, w))
        elif k % 4 == 1:
            v = (1500 + 900 * np.sin(xx / 4.0) * np.cos(yy / 6.0) + rng.normal(0, 15, (h, w))).clip(1)
        elif k % 4 == 2:
            v = np.where((xx // 5 + yy // 3) % 2 == 0, 300, 2500)
        else:
            v = rng.integers(0, 1 << 16, (h, w))
        rois.append(dict(x=xx[m].astype(np.uint16), y=yy[m].astype(np.uint16), inten=np.asarray(v)[m].astype(np.uint32)))
    b = _abi.batch_from_rois(rois)
    for s in (_abi.default_settings(8), _bank8(_abi.default_settings(8))):
        G = hip_ctx.featurize_host(b, _abi.FAM_GABOR, s)
        O = po.oracle_featurize(b, _abi.FAM_GABOR, s)
        assert ((G == O) | (np.isnan(G) & np.isnan(O))).all(), np.argwhere(~((G == O) | (np.isnan(G) & np.isnan(O))))[:5]


@pytest.mark.parametrize("top", [255, (1 << 24) - 1, 1 << 24, (1 << 32) - 1])
def test_gabor_default_bank_box_filter_and_zero_rows_exact(hip_ctx, top):
    """The default bank's first filter has f0 = 0 (gabor.cpp:19-25 consumed as :107-110): every tap is (2^-8, 0).  The tiled
    kernel takes its response from exact integer window sums while the ROI's intensities stay below 2^24 and skips the
    all-zero imaginary rows otherwise -- both must give the reference's count ratios bit for bit, on either side of the
    boundary and for flat blocks (ties at the threshold)."""
    rng = np.random.default_rng(top & 0xFFFF)
    rois = []
    for k, (w, h) in enumerate([(61, 61), (17, 9), (64, 30), (8, 40), (33, 33)]):
        yy, xx = np.mgrid[0:h, 0:w]
        m = ((xx - w / 2) ** 2 / (w / 2) ** 2 + (yy - h / 2) ** 2 / (h / 2) ** 2 <= 1.0) if k % 2 == 0 else np.ones((h, w), bool)
        v = rng.integers(max(1, top // 2), top + 1, (h, w), dtype=np.uint64).astype(np.uint32)
        if k == 4:
            v[:, : w // 2] = top                      # a flat block: thousands of equal energies
        v[0, 0] = top
        rois.append(dict(x=xx[m].astype(np.uint16), y=yy[m].astype(np.uint16), inten=v[m]))
    b = _abi.batch_from_rois(rois)
    s = _abi.default_settings(8)
    G = hip_ctx.featurize_host(b, _abi.FAM_GABOR, s)
    O = po.oracle_featurize(b, _abi.FAM_GABOR, s)
    assert ((G == O) | (np.isnan(G) & np.isnan(O))).all()
    if po.have_ref():
        Rf = po.ref_featurize(b, _abi.FAM_GABOR, s, 2)
        assert ((G == Rf) | (np.isnan(G) & np.isnan(Rf))).all()


@pytest.mark.parametrize("n", [9, 20])
def test_gabor_other_kernel_sizes_exact(hip_ctx, n):
    """Bank sizes other than 16 take the generic Gabor kernel."""
    b = _abi.batch_from_rois(_edge_shape_rois(seed=32))
    s = _abi.default_settings(8)
    s.gabor_kersize = n
    G = hip_ctx.featurize_host(b, _abi.FAM_GABOR, s)
    O = po.oracle_featurize(b, _abi.FAM_GABOR, s)
    assert ((G == O) | (np.isnan(G) & np.isnan(O))).all()


@pytest.mark.parametrize("fam", [_abi.FAM_GLRLM, _abi.FAM_GLSZM, _abi.FAM_NGTDM, _abi.FAM_GLRLM | _abi.FAM_GLSZM | _abi.FAM_NGTDM])
@pytest.mark.parametrize("gd", [8, -20])
def test_texture_families_alone_on_spilled_rois(hip_ctx, fam, gd):
    """Each texture family ALONE over a batch that mixes LDS-sized ROIs with ROIs that take the global workspace: the carve-out
    (hence the register tier of roi_texture_kernel) depends on the family set, so the combinations are not covered by the
    all-families tests.  GLRLM alone used to return zero rows for the spilled ROIs."""
    b = _abi.batch_from_rois(_large_rois())
    s = _abi.default_settings(gd)
    G = hip_ctx.featurize_host(b, fam, s)
    O = po.oracle_featurize(b, fam, s)
    assert not parity.compare_tables(G, O, _lib.column_names(fam, s))


@pytest.mark.parametrize("gd", [17, 20, 32, 48, -8, -24])
def test_glcm_in_kernel_features_at_tight_register_tiers(hip_ctx, gd):
    """Grey depths above 16 (and radiomics binning) keep the GLCM features inside roi_features_kernel; with small ROIs the
    carve-out is small enough for the 72/80/96-register builds."""
    rois = synth.random_rois(120, seed=77, rmax=14)
    b = _abi.batch_from_rois(rois)
    s = _abi.default_settings(gd)
    mask = _abi.FAM_INTENSITY | _abi.FAM_GLCM
    G = hip_ctx.featurize_host(b, mask, s)
    O = po.oracle_featurize(b, mask, s)
    assert not parity.compare_tables(G, O, _lib.column_names(mask, s))


@pytest.mark.parametrize("mask", [_abi.FAM_INTENSITY | _abi.FAM_GLCM, _abi.FAM_GLCM | _abi.FAM_GLSZM | _abi.FAM_GLDM])
def test_deep_glcm_with_a_spilled_roi(hip_ctx, mask):
    """A grey depth whose co-occurrence matrix does not fit LDS (200 levels) in a batch that ALSO holds a ROI for the global
    workspace: the INTENSITY + GLCM group must fall back to the workspace in the spill flow too (it used to refuse)."""
    rng = np.random.default_rng(5)
    ys, xs = np.nonzero(np.ones((300, 280), bool))
    rois = synth.random_rois(10, seed=3, rmax=14) + [dict(x=xs, y=ys, inten=rng.integers(1, 4096, len(xs)).astype(np.uint32))]
    b = _abi.batch_from_rois(rois)
    s = _abi.default_settings(200)
    G = hip_ctx.featurize_host(b, mask, s)
    O = po.oracle_featurize(b, mask, s)
    assert not parity.compare_tables(G, O, _lib.column_names(mask, s))


@pytest.mark.parametrize("gd", [3, 8])
def test_texture_row_scans_at_edge_widths(hip_ctx, gd):
    """GLRLM (one wave per direction, run state shifted along the diagonals), GLSZM (row sweep) and NGTDM on boxes of width 1, 2,
    63 and 64 -- the widths at which a diagonal run leaves the wave -- with few grey levels, so that long runs exist."""
    rng = np.random.default_rng(17)
    rois = []
    for (w, h) in [(1, 1), (1, 30), (30, 1), (2, 2), (2, 40), (63, 5), (64, 5), (63, 63), (64, 64), (64, 1), (5, 64), (40, 64)]:
        m = np.ones((h, w), bool)
        if w > 3 and h > 3:
            m &= rng.random((h, w)) > 0.1
            m[0, 0] = m[0, w - 1] = m[h - 1, 0] = m[h - 1, w - 1] = True
        ys, xs = np.nonzero(m)
        v = rng.integers(1, 4, len(xs)) * 1000                    # three well-separated intensities -> long runs under any binning
        rois.append(dict(x=xs, y=ys, inten=v.astype(np.uint32)))
    b = _abi.batch_from_rois(rois)
    s = _abi.default_settings(gd)
    G = hip_ctx.featurize_host(b, TEX, s)
    O = po.oracle_featurize(b, TEX, s)
    assert not parity.compare_tables(G, O, _lib.column_names(TEX, s))
    if po.have_ref():
        assert not parity.compare_tables(G, po.ref_featurize(b, TEX, s, 2), _lib.column_names(TEX, s))


@pytest.mark.parametrize("gd", [3, 8, 40])
def test_texture_families_on_workspace_rois_wider_than_four_chunks(hip_ctx, gd):
    """ROIs beyond LDS run from the global workspace, whose build keeps the register sweeps for boxes up to 512 wide (eight
    64-column chunks), counts runs of up to 16 pixels in an LDS copy of the matrix columns and keeps the NGTDM accumulators in
    LDS: widths on both sides of 256 and 512, few grey levels, whole rows of one level (runs longer than the LDS columns and
    longer than four chunks), a level that appears once."""
    rng = np.random.default_rng(31)
    rois = []
    for (w, h) in [(300, 280), (257, 260), (512, 140), (513, 130), (448, 150)]:
        m = rng.random((h, w)) > 0.05
        m[0, 0] = m[0, w - 1] = m[h - 1, 0] = m[h - 1, w - 1] = True
        ys, xs = np.nonzero(m)
        v = rng.integers(1, 5, len(xs)) * 900
        v[ys == 7] = 2700                                        # a run across every chunk of the row
        v[(ys == 9) & (xs >= 250) & (xs < 270)] = 1800           # 20 pixels: beyond the LDS columns, across the 256 boundary
        v[(ys == 11) & (xs < 17)] = 900                          # 17 pixels at the left edge
        v[(ys % 50 == 20) & (xs % 64 == 63)] = 3600              # single pixels on chunk ends
        v[len(v) // 2] = 4000                                    # a level of its own
        rois.append(dict(x=xs, y=ys, inten=v.astype(np.uint32)))
    b = _abi.batch_from_rois(rois)
    s = _abi.default_settings(gd)
    G = hip_ctx.featurize_host(b, TEX, s)
    assert any(r["workspace"] for r in hip_ctx.launch_report())
    assert not parity.compare_tables(G, po.oracle_featurize(b, TEX, s), _lib.column_names(TEX, s))
    for fam in (_abi.FAM_GLRLM, _abi.FAM_NGTDM):                 # alone: other LDS offsets of the hybrid state
        G1 = hip_ctx.featurize_host(b, fam, s)
        assert not parity.compare_tables(G1, po.oracle_featurize(b, fam, s), _lib.column_names(fam, s))


@pytest.mark.parametrize("gd", [3, 8])
def test_texture_row_scans_on_boxes_wider_than_a_wave(hip_ctx, gd):
    """Boxes 65 .. 128 wide take the two-chunk register sweeps (GLSZM: labels and runs crossing column 63 | 64 through
    v_readlane), 129 and up the chunked LDS sweep: widths on both sides of both boundaries, few grey levels (long runs and
    large zones that straddle the chunk boundary), holes, a zone that snakes across the boundary several times."""
    rng = np.random.default_rng(23)
    rois = []
    for (w, h) in [(65, 5), (65, 65), (66, 3), (96, 40), (127, 9), (128, 20), (128, 1), (129, 6), (70, 70), (100, 2), (30, 100), (128, 128),
                   (129, 20), (192, 9), (193, 12), (255, 7), (256, 20), (257, 5), (200, 60)]:
        m = np.ones((h, w), bool)
        if h > 3:
            m &= rng.random((h, w)) > 0.08
            m[0, 0] = m[0, w - 1] = m[h - 1, 0] = m[h - 1, w - 1] = True
        ys, xs = np.nonzero(m)
        v = rng.integers(1, 4, len(xs)) * 1000
        if h >= 9:                                               # a one-level snake over columns 60..68, rows 0..8
            snake = (xs >= 60) & (xs <= 68) & (ys < 9) & (((ys % 2 == 0)) | ((ys % 4 == 1) & (xs == 68)) | ((ys % 4 == 3) & (xs == 60)))
            v[snake] = 2000
        if h == 20 and w >= 200:                                 # one level across all four chunks, and runs that end / begin at each boundary
            v[ys == 11] = 3000
            for cb in (64, 128, 192):
                v[(ys == 13) & (xs >= cb - 3) & (xs <= cb + 2)] = 2000
                v[(ys == 15) & (xs == cb - 1)] = 1000
                v[(ys == 15) & (xs == cb)] = 2000
        if h == 20:                                              # whole rows of one level: runs of 128, 65, 64 across the boundary
            v[ys == 3] = 3000
            v[(ys == 5) & (xs <= 64)] = 3000
            v[(ys == 7) & (xs >= 64)] = 3000
            v[(ys == 9) & (xs <= 63)] = 3000
        rois.append(dict(x=xs, y=ys, inten=v.astype(np.uint32)))
    b = _abi.batch_from_rois(rois)
    s = _abi.default_settings(gd)
    G = hip_ctx.featurize_host(b, TEX, s)
    O = po.oracle_featurize(b, TEX, s)
    assert not parity.compare_tables(G, O, _lib.column_names(TEX, s))
    if po.have_ref():
        assert not parity.compare_tables(G, po.ref_featurize(b, TEX, s, 2), _lib.column_names(TEX, s))
    for fam in (_abi.FAM_GLSZM, _abi.FAM_GLRLM, _abi.FAM_NGTDM):  # each family alone (its own launch layout)
        G1 = hip_ctx.featurize_host(b, fam, s)
        assert not parity.compare_tables(G1, po.oracle_featurize(b, fam, s), _lib.column_names(fam, s))


@pytest.mark.parametrize("gd", [8, 16, 64])
def test_intensity_glcm_on_boxes_wider_than_a_wave(hip_ctx, gd):
    """INTENSITY + GLCM on boxes 65 .. 129 wide: the co-occurrence sweep takes two columns per lane up to 128 (the E / SE / SW
    pairs across column 63 | 64 through v_readlane) and the per-pixel loop beyond; pairs that straddle the boundary, zero
    intensities (skipped by the scan) on both sides of it, last rows and last columns."""
    rng = np.random.default_rng(29)
    rois = []
    for (w, h) in [(65, 4), (65, 65), (66, 1), (100, 30), (127, 9), (128, 16), (129, 6), (71, 71)]:
        m = np.ones((h, w), bool)
        if h > 3:
            m &= rng.random((h, w)) > 0.05
            m[0, 0] = m[0, w - 1] = m[h - 1, 0] = m[h - 1, w - 1] = True
        ys, xs = np.nonzero(m)
        v = rng.integers(0, 4096, len(xs))
        v[(xs >= 62) & (xs <= 66) & (ys % 3 == 0)] = 0            # zero intensities around the chunk boundary
        v[0] = 4095
        rois.append(dict(x=xs, y=ys, inten=v.astype(np.uint32)))
    b = _abi.batch_from_rois(rois)
    s = _abi.default_settings(gd)
    mask = _abi.FAM_INTENSITY | _abi.FAM_GLCM
    G = hip_ctx.featurize_host(b, mask, s)
    O = po.oracle_featurize(b, mask, s)
    assert not parity.compare_tables(G, O, _lib.column_names(mask, s))
    if po.have_ref():
        assert not parity.compare_tables(G, po.ref_featurize(b, mask, s, 2), _lib.column_names(mask, s))


def test_output_is_bit_reproducible(hip_ctx):
    """Twenty repeated calls of all twelve families: identical bits every time (the GLSZM cell table is an ordered hash, every
    other accumulation is integer or runs in a fixed order) -- a first-come hash made GLSZM_ZE / GLV / SALGLE move by 1-2 ulp."""
    b = _abi.batch_from_rois(synth.random_rois(120, seed=5, rmax=25))
    s = _abi.default_settings(8)
    first = hip_ctx.featurize_host(b, _abi.FAM_ALL, s)
    for _ in range(20):
        again = hip_ctx.featurize_host(b, _abi.FAM_ALL, s)
        assert ((again == first) | (np.isnan(again) & np.isnan(first))).all()


def test_gabor_fused_variant_matches_on_noisy_fields():
    """NYXHIP_GABOR_FUSED=1 (one FMA per tap) in a child process: on noisy intensity fields the count ratios equal the default
    (exact) kernel's.  (On tie-laden flat fields they may not: tools/gabor_fuzz.py; that is why the variant is opt-in.)"""
    import os, subprocess, sys, tempfile
    code = ("import sys, numpy as np\n"
            "sys.path.insert(0, '.')\n"
            "from nyxus_amd import _abi, _lib\n"
            "from tests import synth\n"
            "b = _abi.batch_from_rois(synth.random_rois(150, seed=9, rmax=28))\n"
            "np.save(sys.argv[1], _lib.Context(0).featurize_host(b, _abi.FAM_GABOR, _abi.default_settings(8)))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        out = {}
        for tag, env in (("exact", {}), ("fused", {"NYXHIP_GABOR_FUSED": "1"})):
            path = os.path.join(d, tag + ".npy")
            subprocess.run([sys.executable, "-c", code, path], check=True, cwd=root, env=dict(os.environ, **env))
            out[tag] = np.load(path)
    assert np.array_equal(out["exact"], out["fused"])


# ---- the reference's default grey depth (17..64 matlab levels): 16-bit matrices + marginal-based features (roi_features_kernel_g16) ----
@pytest.mark.parametrize("irregular", [False, True])
@pytest.mark.parametrize("gd", [64, 33, 17])
def test_benchmark_tile_default_grey_depth(hip_ctx, gd, irregular):
    b = synth.tile_batch(2, irregular=irregular)
    s = _abi.default_settings(gd)
    G = _check(hip_ctx, b, MASK, s)
    assert G.shape == (196, 185)


@pytest.mark.parametrize("gd,na,sym,offset,mask", [(64, 4, 0, 1, MASK), (40, 3, 1, 1, MASK), (64, 2, 0, 2, MASK), (17, 1, 1, 3, _abi.FAM_GLCM),
                                                   (64, 4, 0, 1, _abi.FAM_GLCM), (50, 4, 1, 1, _abi.FAM_GLCM)])
def test_default_grey_depth_variants(hip_ctx, gd, na, sym, offset, mask):
    """Small-range ROIs (so that the 16-bit launch is the one taken), angle subsets, symmetric matrices, offsets beyond 1 (the
    generic sweep of the 16-bit path), GLCM alone; zero-valued pixels, constant and blank ROIs come with random_rois."""
    rois = synth.random_rois(90, seed=23, value_modes=(4096, 256, 8, 1000))
    rois.append(dict(x=list(range(70)) * 3, y=[0] * 70 + [1] * 70 + [2] * 70, inten=list(np.random.default_rng(1).integers(1, 500, 210))))   # wider than one wave
    s = _abi.default_settings(gd)
    s.glcm_n_angles = na
    for i, a in enumerate((0, 45, 90, 135)[4 - na:]):
        s.glcm_angles[i] = a
    s.glcm_symmetric = sym
    s.glcm_offset = offset
    _check(hip_ctx, _abi.batch_from_rois(rois), mask, s)


# ---- round-2 code paths that the default settings do not reach ------------------------------------------------------------------
TEX_ALL = _abi.FAM_GLRLM | _abi.FAM_GLSZM | _abi.FAM_NGTDM
DEP_ALL = _abi.FAM_GLDZM | _abi.FAM_GLDM | _abi.FAM_NGLDM


def test_texture_sixteen_bit_plane(hip_ctx):
    """A grey depth above 254 keeps the texture kernel on its 16-bit plane (the 8-bit plane / eight-workgroup build serves <= 254)."""
    rois = synth.random_rois(40, seed=21)
    _check(hip_ctx, _abi.batch_from_rois(rois), TEX_ALL, _abi.default_settings(300), against_ref=False)


@pytest.mark.parametrize("gd", [8, 63, 64])
def test_dependence_tails_parallel_equals_sequential(hip_ctx, gd, monkeypatch):
    """The three feature tails side by side on three waves (default) against one after the other (NYXHIP_DEP_SEQ=1): the same bits.
    gd 63 is the last depth served by the byte planes, 64 the first on the 16-bit ones."""
    b = _abi.batch_from_rois(synth.random_rois(60, seed=22))
    s = _abi.default_settings(gd)
    par = hip_ctx.featurize_host(b, DEP_ALL, s)
    monkeypatch.setenv("NYXHIP_DEP_SEQ", "1")
    seq = hip_ctx.featurize_host(b, DEP_ALL, s)
    assert np.array_equal(par.view(np.uint64), seq.view(np.uint64))
    monkeypatch.delenv("NYXHIP_DEP_SEQ")
    O = po.oracle_featurize(b, DEP_ALL, s)
    bad = parity.compare_tables(par, O, _lib.column_names(DEP_ALL, s))
    assert not bad, "\n".join(bad[:20])


def test_grey_depth_64_split_launch_equals_fused(hip_ctx, monkeypatch):
    """INTENSITY + GLCM at 17..64 levels run as two launches (intensity at eight workgroups per CU); NYXHIP_G16_FUSED=1 keeps the one
    launch: the same bits either way."""
    b = synth.tile_batch(2, irregular=True)
    s = _abi.default_settings(64)
    split = hip_ctx.featurize_host(b, MASK, s)
    monkeypatch.setenv("NYXHIP_G16_FUSED", "1")
    fused = hip_ctx.featurize_host(b, MASK, s)
    # (round 6: in the split form the INTENSITY columns of the smallest size class come from the wave-per-ROI kernel -- another fixed
    #  summation order -- so those rows agree within the parity tolerance, everybody else bit for bit)
    n_px = np.diff(np.asarray(b.px_offset).astype(np.int64))
    side = np.maximum(np.asarray(b.bbox_w), np.asarray(b.bbox_h))
    small = (n_px <= 256) & (side <= 32)
    assert np.array_equal(split[~small].view(np.uint64), fused[~small].view(np.uint64))
    assert small.any() and not parity.compare_tables(split[small], fused[small], _lib.column_names(MASK, s))


def test_contour_wide_boxes_take_the_position_wise_pass(hip_ctx):
    """Padded planes wider than a wave (w + 2 > 64) keep the position-wise candidate / X-crossing passes of the contour kernel."""
    yy, xx = np.mgrid[0:50, 0:90]
    m = ((xx - 45) / 44.0) ** 2 + ((yy - 25) / 24.0) ** 2 <= 1.0
    m[20:30, 40:50] = False                                  # a hole: a second contour
    ys, xs = np.nonzero(m)
    rng = np.random.default_rng(5)
    roi = {"x": xs.astype(np.int64), "y": ys.astype(np.int64), "inten": rng.integers(1, 4096, xs.size).astype(np.uint32)}
    small = synth.random_rois(6, seed=23)
    mask = _abi.FAM_SMOMS | _abi.FAM_IMOMS
    _check(hip_ctx, _abi.batch_from_rois([roi] + small), mask, _abi.default_settings(8), against_ref=False)


@pytest.mark.gpu
def test_eight_wave_64_level_launch_gives_the_bits_of_the_four_wave_one():
    """Grey depths 17..64: classes whose largest box has >= 48 x 48 cells run the GLCM launch on eight waves per ROI with the feature pass on
    two waves per angle (roi_features_kernel_g16w8); which launch a ROI meets depends on its companions' boxes, so both must give the same
    bits.  Two child processes (the A/B knob NYXHIP_G16_W4 is read once per process) over mixed batches, three depths, both symmetries."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "g16_w8_check.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-1500:]
