"""GPU tests of the Python face: `nyxus_amd.Nyxus.featurize` against input/output pairs captured from the
reference's own `Nyxus.featurize()` (tests/golden/api_featurize.json, made by tests/golden/make_api_golden.py)."""
import json
import os

import numpy as np
import pytest

import nyxus_amd
from tests import parity

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "api_featurize.json")))


@pytest.mark.parametrize("case", sorted(GOLD))
def test_featurize_matches_reference_dataframe(case):
    g = GOLD[case]
    inten = np.array(g["inten"], dtype=np.dtype(g["inten_dtype"]))
    seg = np.array(g["seg"])
    nyx = nyxus_amd.Nyxus(g["features"], **g["kwargs"])
    for mp in g.get("metaparams", []):
        nyx.set_metaparam(mp)
    if g["names"]:
        df = nyx.featurize(inten, seg, intensity_names=g["names"][0], label_names=g["names"][1])
    else:
        df = nyx.featurize(inten, seg)
    assert list(df.columns) == g["columns"]                      # header: names, order, angle expansion
    str_cols = [c for c in g["columns"] if c not in g["numeric_columns"]]
    assert df[str_cols].values.tolist() == g["strings"]          # image / mask names, row order
    assert df["ROI_label"].dtype == np.uint32
    got = df[g["numeric_columns"]].values.astype(float)
    want = np.array(g["numeric"], dtype=float)
    exact = parity.EXACT_COLUMNS | {"ROI_label", "t_index"} | {c for c in g["numeric_columns"] if c.startswith("GABOR_")}
    # first-order central moments are zero up to cancellation noise: bound them by 1e-9 * m00 * image side
    atol = {}
    cols = g["numeric_columns"]
    for cm, m00 in (("CENTRAL_MOMENT_01", "SPAT_MOMENT_00"), ("CENTRAL_MOMENT_10", "SPAT_MOMENT_00"), ("IMOM_CM_01", "IMOM_RM_00"), ("IMOM_CM_10", "IMOM_RM_00")):
        if cm in cols and m00 in cols:
            atol[cm] = 1e-9 * np.abs(want[:, cols.index(m00)]) * max(seg.shape[-2:])
    bad = parity.compare_tables(got, want, cols, exact=exact, atol=atol)
    assert not bad, "\n".join(bad[:20])


def test_featurize_validation_messages():
    nyx = nyxus_amd.Nyxus(["MEAN"])
    with pytest.raises(ValueError, match="must be numpy.ndarray"):
        nyx.featurize([[1]], np.zeros((1, 1)))
    with pytest.raises(ValueError, match="same dimension"):
        nyx.featurize(np.zeros((2, 2)), np.zeros((1, 2, 2)))
    with pytest.raises(ValueError, match="matching dimensions"):
        nyx.featurize(np.zeros((2, 2)), np.zeros((2, 3)))
    with pytest.raises(ValueError, match="Invalid output type"):
        nyx.featurize(np.zeros((2, 2)), np.zeros((2, 2)), output_type="csv")
    with pytest.raises(ValueError, match="not served by the MI355X path"):
        nyxus_amd.Nyxus(["PERIMETER"])
    with pytest.raises(ValueError, match="non-negative"):
        nyxus_amd.Nyxus(["MEAN"], coarse_gray_depth=0)


DIRGOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "api_directory.json")))


@pytest.mark.parametrize("case", sorted(DIRGOLD))
def test_featurize_directory_matches_reference(case):
    """The reference's own OME-TIFF fixture (139 ROIs over two channel pairs, tests/python/data) end to end:
    TIFF ingest -> slide prescan -> ROI assembly -> HIP reduce -> DataFrame, vs the reference's DataFrame."""
    g = DIRGOLD[case]
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiff")
    nyx = nyxus_amd.Nyxus(g["features"], **g["kwargs"])
    if "files" in g:      # featurize_files (reference nyxus.py:524-593) on the same fixture
        df = nyx.featurize_files([os.path.join(root, "int", f) for f in g["files"]], [os.path.join(root, "seg", f) for f in g["files"]], False)
    else:
        df = nyx.featurize_directory(os.path.join(root, "int"), os.path.join(root, "seg"))
    assert list(df.columns) == g["columns"]
    str_cols = [c for c in g["columns"] if c not in g["numeric_columns"]]
    assert df[str_cols].values.tolist() == g["strings"]
    got = df[g["numeric_columns"]].values.astype(float)
    want = np.array(g["numeric"], dtype=float)
    assert got.shape == want.shape == (139, len(g["numeric_columns"]))
    cols = g["numeric_columns"]
    atol = {cm: 1e-9 * np.abs(want[:, cols.index(m00)]) * 1024 for cm, m00 in (("CENTRAL_MOMENT_01", "SPAT_MOMENT_00"), ("CENTRAL_MOMENT_10", "SPAT_MOMENT_00"))
            if cm in cols and m00 in cols}
    bad = parity.compare_tables(got, want, cols, exact=parity.EXACT_COLUMNS | {"ROI_label", "t_index"}, atol=atol)
    assert not bad, "\n".join(bad[:20])


def test_featurize_directory_on_a_tiled_4096_slide(tmp_path):
    """A 4096 x 4096 slide stored as 1024 x 1024 deflate tiles (written from the seeded recipe of tests/synth.py), ROIs crossing
    the TIFF tile borders, sparse label values up to 256000, ROIs of up to 38 k pixels: native tile-by-tile decode ->
    device scan / prescan / reduce, against the reference package's DataFrame (tests/golden/api_directory_tiled.json)."""
    from tests import synth
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "api_directory_tiled.json")))["slide4096_tiled_gd8"]
    os.makedirs(tmp_path / "int"); os.makedirs(tmp_path / "seg")
    I, L = synth.slide4096(g["slide_seed"])
    synth.write_tiled_tiff(str(tmp_path / "int" / "slide0.ome.tif"), I)
    synth.write_tiled_tiff(str(tmp_path / "seg" / "slide0.ome.tif"), L)
    df = nyxus_amd.Nyxus(g["features"], **g["kwargs"]).featurize_directory(str(tmp_path / "int"), str(tmp_path / "seg"))
    assert list(df.columns) == g["columns"]
    str_cols = [c for c in g["columns"] if c not in g["numeric_columns"]]
    assert df[str_cols].values.tolist() == g["strings"]
    got = df[g["numeric_columns"]].values.astype(float)
    want = np.array(g["numeric"], dtype=float)
    assert got.shape == want.shape == (256, len(g["numeric_columns"]))
    bad = parity.compare_tables(got, want, g["numeric_columns"], exact=parity.EXACT_COLUMNS | {"ROI_label", "t_index"})
    assert not bad, "\n".join(bad[:20])
