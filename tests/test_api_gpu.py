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


def _moment_floors(images, masks, want, cols):
    """Central moments and what is derived from them (normalised, Hu) are sums with cancellation: floors derived from the size of
    the cancelling terms (tests/parity.py: moment_atol), per ROI, from the same pixels the call saw.  None without such columns."""
    if not any(c.startswith(("CENTRAL_MOMENT_", "IMOM_CM_")) for c in cols):
        return None
    from tests import roi_assembly
    r0, atol = 0, {}
    for it, sg in zip(images, masks):
        it = np.asarray(it).astype(np.float64)
        if it.min() < 0:
            it = it - it.min()                                      # the package's shift of negative intensities (nyxus.py:480-489)
        b = roi_assembly.assemble(it.astype(np.uint32), np.asarray(sg).astype(np.uint32), 1.7976931348623157e308, -1.7976931348623157e308)
        if b is None:
            continue
        for k, v in parity.moment_atol(b, want[r0:r0 + b.n_roi], cols).items():
            atol.setdefault(k, []).append(v)
        r0 += b.n_roi
    assert r0 == want.shape[0]
    return {k: np.concatenate(v) for k, v in atol.items()}


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
    cols = g["numeric_columns"]
    atol = _moment_floors(inten if inten.ndim == 3 else inten[None], seg if seg.ndim == 3 else seg[None], want, cols)
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
    from nyxus_amd import tiff_ingest
    names = g["files"] if "files" in g else sorted(os.listdir(os.path.join(root, "int")))
    atol = _moment_floors([tiff_ingest.read_tiff(os.path.join(root, "int", f)) for f in names],
                          [tiff_ingest.read_tiff(os.path.join(root, "seg", f)) for f in names], want, cols)
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
