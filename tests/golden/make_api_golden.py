#!/usr/bin/env python3
"""Generates tests/golden/api_featurize.json: input/output pairs of the reference's own
`Nyxus.featurize()` (the Python API on top of the hot path), captured by importing the reference
package in the build container.  Only DATA is stored: input arrays + constructor kwargs + the
resulting DataFrame (columns, string columns, numeric values).

The reference Python package needs its compiled pybind11 backend.  It is looked up in
$NYXUS_REF_PKG (default /tmp/nyx_oracle_pkg, built from /root/reference by the recipe in
SURVEY.md section 8c) and must be run with the system libstdc++ preloaded:

    LD_PRELOAD=/usr/lib/x86_64-linux-gnu/libstdc++.so.6 python tests/golden/make_api_golden.py
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("NYXUS_REF_PKG", "/tmp/nyx_oracle_pkg"))
import nyxus  # noqa: E402  (the REFERENCE package)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "api_featurize.json")


def run(features, kwargs, inten, seg, names=None, metaparams=None):
    nyx = nyxus.Nyxus(features, **kwargs)
    for mp in metaparams or []:
        nyx.set_metaparam(mp)
    if names:
        df = nyx.featurize(inten.copy(), seg.copy(), intensity_names=names[0], label_names=names[1])
    else:
        df = nyx.featurize(inten.copy(), seg.copy())
    num = df.select_dtypes(include=[np.number])
    return {"features": features, "kwargs": kwargs, "inten": inten.tolist(), "inten_dtype": str(inten.dtype),
            "seg": seg.tolist(), "names": names, "metaparams": metaparams or [], "columns": list(df.columns),
            "strings": df[[c for c in df.columns if c not in num.columns]].values.tolist(),
            "numeric_columns": list(num.columns), "numeric": num.values.astype(float).tolist()}


def main():
    cases = {}
    # (1) README pair (/root/reference/README.md:97-109)
    I = np.array([[1, 4, 4, 1, 1], [1, 4, 6, 1, 1], [4, 1, 6, 4, 1], [4, 4, 6, 4, 1]])
    S = np.array([[1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 1, 1, 1, 1], [1, 1, 1, 1, 1]])
    cases["readme_mean_median"] = run(["MEAN", "MEDIAN"], {}, I, S)
    cases["readme_all_intensity"] = run(["*ALL_INTENSITY*"], {}, I, S)
    cases["readme_all_glcm_gd8"] = run(["*ALL_GLCM*"], {"coarse_gray_depth": 8}, I, S)
    # (2) seeded 48x48 multi-ROI tile, 3 images, every hot-path family; holes, a constant ROI, a zero-valued pixel
    rng = np.random.default_rng(7)
    inten = rng.integers(1, 4096, (3, 48, 48)).astype(np.uint32)
    seg = np.zeros((3, 48, 48), np.uint32)
    yy, xx = np.mgrid[0:48, 0:48]
    k = 0
    for (cy, cx, r) in [(10, 10, 8), (12, 34, 9), (35, 12, 10), (34, 35, 7), (24, 24, 3)]:
        k += 1
        seg[:, ((yy - cy) ** 2 + (xx - cx) ** 2) <= r * r] = k
    seg[1][(yy + xx) % 7 == 0] = 0          # holes
    inten[2][seg[2] == 5] = 77              # constant ROI
    inten[0, 10, 10] = 0                    # zero-valued pixel inside ROI 1
    fam = ["*ALL_INTENSITY*", "*ALL_GLCM*", "*ALL_GLRLM*", "*ALL_GLSZM*", "*ALL_NGTDM*", "GABOR", "ZERNIKE2D"]
    cases["tile48_all_gd8"] = run(fam, {"coarse_gray_depth": 8}, inten, seg, names=(["i0.tif", "i1.tif", "i2.tif"], ["s0.tif", "s1.tif", "s2.tif"]))
    cases["tile48_all_gd64"] = run(fam, {}, inten[:1], seg[:1])
    cases["tile48_ibsi"] = run(["*ALL_GLCM*", "*ALL_GLRLM*", "*ALL_NGTDM*"], {"ibsi": True}, (inten[:1] % 9 + 1).astype(np.uint32), seg[:1])
    cases["tile48_gabor8"] = run(["GABOR"], {"gabor_freqs": [4, 16, 32, 64, 4, 16, 32, 64], "gabor_thetas": [0, 22.5, 45, 67.5, 90, 112.5, 135, 157.5]}, inten[:1], seg[:1])
    # (2b) the dependence / distance-zone families: matlab binning and the IBSI path
    dep = ["*ALL_GLDZM*", "*ALL_GLDM*", "*ALL_NGLDM*"]
    cases["tile48_dep_gd8"] = run(dep, {"coarse_gray_depth": 8}, inten, seg)
    cases["tile48_dep_gd64_mixed"] = run(["*ALL_GLRLM*", "*ALL_GLSZM*", "*ALL_NGTDM*", "MEAN"] + dep, {}, inten[:1], seg[:1])
    cases["tile48_dep_ibsi"] = run(dep, {"ibsi": True}, (inten[:1] % 9 + 1).astype(np.uint32), seg[:1])
    # (2c) 2-D geometric moments (contour + weighted moments included)
    cases["tile48_geomoms"] = run(["*GEOMOMS*"], {}, inten[:2], seg[:2])
    cases["tile48_sgeomoms_mixed"] = run(["*SGEOMOMS*", "IMOM_HU1", "IMOM_WHU3", "MEAN", "ZERNIKE2D"], {}, inten[:1], seg[:1])
    # (3) negative / float-valued input: shift-and-cast of nyxus.py:480-489
    ct = (rng.normal(0, 300, (1, 32, 32))).astype(np.float32)
    sg = np.zeros((1, 32, 32), np.uint32)
    sg[0, 4:20, 5:25] = 3
    sg[0, 22:30, 2:12] = 9
    cases["hounsfield_float"] = run(["*ALL_INTENSITY*"], {}, ct, sg)
    # (4) SURVEY 8(b) fixture (2): the 4-slice 4x5 IBSI phantom of the reference's own API test
    # (/root/reference/tests/python/test_data.py:3-49; call shape and asserted values /root/reference/tests/python/test_nyxus.py:410-447)
    ph_i = np.array([[[1, 4, 4, 1, 1], [1, 4, 6, 1, 1], [4, 1, 6, 4, 1], [4, 4, 6, 4, 1]],
                     [[1, 4, 4, 1, 1], [1, 1, 6, 1, 1], [1, 1, 3, 1, 1], [4, 4, 6, 1, 1]],
                     [[1, 4, 4, 1, 1], [1, 1, 1, 1, 1], [1, 1, 6, 1, 1], [1, 1, 6, 1, 1]],
                     [[1, 4, 4, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 6, 1, 1]]])
    ph_s = np.array([[[1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1]],
                     [[1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 1, 1, 1, 1], [1, 1, 1, 1, 1]],
                     [[1, 1, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 0, 1, 1], [1, 1, 1, 1, 1]],
                     [[1, 1, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1]]])
    ibsi16 = ["GLCM_ASM", "GLCM_CONTRAST", "GLCM_CORRELATION", "GLCM_DIFAVE", "GLCM_DIFENTRO", "GLCM_DIFVAR", "GLCM_ENERGY", "GLCM_ENTROPY",
              "GLCM_HOM1", "GLCM_INFOMEAS1", "GLCM_INFOMEAS2", "GLCM_IDM", "GLCM_SUMAVERAGE", "GLCM_SUMENTROPY", "GLCM_SUMVARIANCE", "GLCM_VARIANCE"]
    nm = ["test_name_1", "test_name_2", "test_name_3", "test_name_4"]
    cases["ibsi_phantom_3d"] = run(ibsi16, {"ibsi": True}, ph_i, ph_s, names=(nm, nm), metaparams=["glcm/greydepth=100", "glcm/offset=1"])
    cases["ibsi_phantom_2d_all_glcm"] = run(["*ALL_GLCM*"], {"ibsi": True}, ph_i[0], ph_s[0], names=(nm[:1], nm[:1]))
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    if only:                                   # add / refresh the named cases, keep the others as captured
        old = json.load(open(OUT))
        old.update({k: cases[k] for k in only})
        cases = old
    with open(OUT, "w") as fh:
        json.dump(cases, fh, separators=(",", ":"))
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", {k: (len(v["numeric"]), len(v["columns"])) for k, v in cases.items()})


if __name__ == "__main__":
    main()
