"""CPU tests of the boundary: the C-ABI library loads and exports every symbol
include/nyxhip.h declares; column catalogue; argument validation that needs no GPU."""
import ctypes as C
import os
import re

import pytest

from nyxus_amd import _abi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "nyxhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(nyxhip_[a-z_0-9]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/nyxhip.h but not exported"
    assert sorted(_lib.ABI_SYMBOLS) == declared
    assert lib.nyxhip_abi_version() == 2


def test_struct_layout_matches_header_defaults():
    """nyxhip_default_settings (C side) and _abi.default_settings (ctypes mirror) agree
    field by field -- catches struct-layout drift between the header and the mirror."""
    lib = _lib.load()
    a = _abi.Settings()
    lib.nyxhip_default_settings(C.byref(a))
    b = _abi.default_settings(64)
    for name, _ in _abi.Settings._fields_:
        va, vb = getattr(a, name), getattr(b, name)
        if hasattr(va, "__len__"):
            assert list(va) == pytest.approx(list(vb)), name
        else:
            assert va == pytest.approx(vb), name


def test_column_catalogue():
    s = _abi.default_settings(8)
    names = _lib.column_names(_abi.FAM_INTENSITY | _abi.FAM_GLCM, s)
    assert len(names) == 36 + 30 * 4 + 29 == 185
    assert names[0] == "COV" and names[35] == "UNIFORMITY_PIU"
    assert names[36:40] == ["GLCM_ASM_0", "GLCM_ASM_45", "GLCM_ASM_90", "GLCM_ASM_135"]
    assert names[36 + 120] == "GLCM_ASM_AVE" and names[-1] == "GLCM_SUMVARIANCE_AVE"
    assert "GLCM_HOM2_AVE" not in names  # featureset.h:205-233 has no HOM2_AVE
    s.glcm_n_angles = 2
    assert len(_lib.column_names(_abi.FAM_GLCM, s)) == 30 * 2 + 29


def test_init_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.NyxHipError) as ei:
        _lib.Context(0)
    assert ei.value.code == 2  # NYXHIP_ERR_NO_DEVICE: no CPU fallback behind the ABI


def test_product_never_imports_oracle():
    """The product package must not reference the test-only checkers."""
    pkg = os.path.join(ROOT, "nyxus_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("no CPU fallback", ""), os.path.join(dp, f)


def test_python_parameter_surface_without_a_gpu():
    """set_params / get_params / set_metaparam / get_metaparam of the reference's Nyxus class (nyxus.py:264-300, :769-868):
    pure bookkeeping, no device needed."""
    import nyxus_amd
    n = nyxus_amd.Nyxus(["*ALL_GLCM*", "MEAN"], coarse_gray_depth=8)
    assert n.get_params("coarse_gray_depth", "ibsi") == {"coarse_gray_depth": 8, "ibsi": False}
    n.set_metaparam("glcm/offset=2")
    assert n.get_metaparam("glcm/offset") == 2.0 and n.get_metaparam("glcm/greydepth") == 8.0
    n.set_params(gabor_gamma=0.2, gabor_thetas=[0, 30], gabor_freqs=[4, 8], ibsi=True, coarse_gray_depth=16, features=["*ALL_GLDM*"])
    p = n.get_params()
    assert p["coarse_gray_depth"] == 16 and p["ibsi"] is True and p["features"] == ["*ALL_GLDM*"]
    assert abs(p["gabor_gamma"] - 0.2) < 1e-7 and [round(t, 4) for t in p["gabor_thetas"]] == [0.0, 30.0] and p["gabor_freqs"] == [4.0, 8.0]
    for bad, exc in ((lambda: n.set_metaparam("glcm/bogus=1"), ValueError), (lambda: n.set_metaparam("glcm/offset"), ValueError),
                     (lambda: n.get_metaparam("glrlm/x"), NameError), (lambda: n.set_params(bogus=1), ValueError),
                     (lambda: n.featurize_files(None, None, False), IOError), (lambda: n.featurize_files([], [], True), ValueError)):
        with pytest.raises(exc):
            bad()
