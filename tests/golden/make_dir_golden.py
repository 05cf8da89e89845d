#!/usr/bin/env python3
"""Generates tests/golden/api_directory.json: output of the reference's `Nyxus.featurize_directory()` on the
reference's own TIFF test data (/root/reference/tests/python/data/{int,seg}; the four data files are
copied verbatim to tests/golden/tiff/ as input fixtures).  Run like make_api_golden.py."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("NYXUS_REF_PKG", "/tmp/nyx_oracle_pkg"))
import nyxus  # noqa: E402  (the REFERENCE package)

HERE = os.path.dirname(os.path.abspath(__file__))
cases = {}
for name, feats, kw in [("intensity_glcm_gd8", ["*ALL_INTENSITY*", "*ALL_GLCM*"], {"coarse_gray_depth": 8}),
                        ("texture_default", ["*ALL_GLRLM*", "*ALL_GLSZM*", "*ALL_NGTDM*", "ZERNIKE2D"], {})]:
    df = nyxus.Nyxus(feats, **kw).featurize_directory(os.path.join(HERE, "tiff", "int"), os.path.join(HERE, "tiff", "seg"))
    num = df.select_dtypes(include=[np.number])
    cases[name] = {"features": feats, "kwargs": kw, "columns": list(df.columns),
                   "strings": df[[c for c in df.columns if c not in num.columns]].values.tolist(),
                   "numeric_columns": list(num.columns), "numeric": num.values.astype(float).tolist()}
    print(name, df.shape)
# featurize_files on the same fixture, the dependence / distance-zone families and the moments (file order as given)
ints = sorted(os.listdir(os.path.join(HERE, "tiff", "int")))
nyx = nyxus.Nyxus(["*ALL_GLDZM*", "*ALL_GLDM*", "*ALL_NGLDM*", "*SGEOMOMS*", "IMOM_HU1"], coarse_gray_depth=16)
df = nyx.featurize_files([os.path.join(HERE, "tiff", "int", f) for f in ints], [os.path.join(HERE, "tiff", "seg", f) for f in ints], False)
num = df.select_dtypes(include=[np.number])
cases["files_dep_moments_gd16"] = {"features": ["*ALL_GLDZM*", "*ALL_GLDM*", "*ALL_NGLDM*", "*SGEOMOMS*", "IMOM_HU1"], "kwargs": {"coarse_gray_depth": 16},
                                   "files": ints, "columns": list(df.columns),
                                   "strings": df[[c for c in df.columns if c not in num.columns]].values.tolist(),
                                   "numeric_columns": list(num.columns), "numeric": num.values.astype(float).tolist()}
print("files", df.shape, nyx.get_params("coarse_gray_depth", "gabor_thetas"))
json.dump(cases, open(os.path.join(HERE, "api_directory.json"), "w"), separators=(",", ":"))

# A synthetic 4096 x 4096 slide stored as a TILED TIFF (1024 x 1024 deflate tiles, ROIs crossing the tile borders, sparse label
# values): the files are written from the seeded recipe of tests/synth.py (the test does the same), only the reference's
# DataFrame is stored -- tests/golden/api_directory_tiled.json.
import tempfile  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import synth  # noqa: E402
tiled = {}
with tempfile.TemporaryDirectory() as tmp:
    os.makedirs(os.path.join(tmp, "int")); os.makedirs(os.path.join(tmp, "seg"))
    I, L = synth.slide4096(0)
    synth.write_tiled_tiff(os.path.join(tmp, "int", "slide0.ome.tif"), I)
    synth.write_tiled_tiff(os.path.join(tmp, "seg", "slide0.ome.tif"), L)
    feats = ["*ALL_INTENSITY*", "GLCM_ASM", "GLCM_CONTRAST", "GLCM_CORRELATION", "GLCM_ENTROPY", "GLCM_INFOMEAS1", "GLCM_SUMAVERAGE"]
    kw = {"coarse_gray_depth": 8}
    df = nyxus.Nyxus(feats, **kw).featurize_directory(os.path.join(tmp, "int"), os.path.join(tmp, "seg"))
    num = df.select_dtypes(include=[np.number])
    tiled["slide4096_tiled_gd8"] = {"features": feats, "kwargs": kw, "slide_seed": 0, "columns": list(df.columns),
                                    "strings": df[[c for c in df.columns if c not in num.columns]].values.tolist(),
                                    "numeric_columns": list(num.columns), "numeric": num.values.astype(float).tolist()}
    print("slide4096", df.shape)
json.dump(tiled, open(os.path.join(HERE, "api_directory_tiled.json"), "w"), separators=(",", ":"))
