#!/usr/bin/env python3
"""Transcribes the DATA held by the reference's own tests for the hot path into
tests/golden/reference_tests.json: pixel fixtures (inputs) and golden-value tables
(expected outputs).  No reference code is copied -- only `{x, y, intensity}` triplets,
image arrays and `{"FEATURE", value}` pairs are parsed out of

  /root/reference/tests/test_data.h            pixelIntensityFeaturesTestData, ibsi_phantom_z*
  /root/reference/tests/test_dsb2018_data.h    4 DSB2018 ROIs
  /root/reference/tests/test_2d_firstorder_{matlab,regression,pyradiomics}.h
  /root/reference/tests/test_2d_glcm_{regression,ibsi}.h
  /root/reference/tests/test_2d_{glrlm,glszm,ngtdm,gldm,ngldm,gldzm}_{regression,ibsi,mirp}.h
  /root/reference/tests/test_gabor_truth.h, test_2d_zernike_regression.h

Run in the build container only (the GPU box has no /root/reference):
    python tests/golden/extract_reference_goldens.py
"""
import json
import os
import re

REF = "/root/reference/tests"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_tests.json")


def strip_comments(src: str) -> str:
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def pixel_arrays(src: str):
    out = {}
    for m in re.finditer(r"NyxusPixel\s+(\w+)\s*\[\]\s*=\s*\{(.*?)\};", src, flags=re.S):
        trip = re.findall(r"\{\s*(\d+)\s*,\s*(\d+)\s*,\s*(\d+)\s*\}", m.group(2))
        out[m.group(1)] = [[int(a), int(b), int(c)] for a, b, c in trip]
    return out


def scalar_maps(src: str):
    out = {}
    for m in re.finditer(r"ref_vals_map<double>\s+(\w+)\s*\{(.*?)\};", src, flags=re.S):
        d = {}
        for k, expr in re.findall(r"\{\s*\"(\w+)\"\s*,\s*([^{}]+?)\}", m.group(2)):
            d[k] = float(eval(expr.strip().rstrip(","), {"__builtins__": {}}))
        out[m.group(1)] = d
    return out


def vector_maps(src: str):
    out = {}
    for m in re.finditer(r"ref_vals_map<std::vector<double>>\s+(\w+)\s*\{(.*?)\};", src, flags=re.S):
        d = {}
        for k, body in re.findall(r"\{\s*\"(\w+)\"\s*,\s*\{(.*?)\}\s*\}", m.group(2), flags=re.S):
            d[k] = [float(x) for x in re.findall(r"[-+0-9.eE]+", body)]
        out[m.group(1)] = d
    return out


def main():
    res = {"_provenance": "parsed from /root/reference/tests/*.h by tests/golden/extract_reference_goldens.py",
           "pixels": {}, "goldens": {}, "vector_goldens": {}}
    td = strip_comments(open(os.path.join(REF, "test_data.h")).read())
    px = pixel_arrays(td)
    for k in ["pixelIntensityFeaturesTestData", "shape2d_morphology_intensity", "shape2d_morphology_mask"] + [f"ibsi_phantom_z{z}_{kind}" for z in range(1, 5) for kind in ("intensity", "mask")]:
        res["pixels"][k] = px[k]
    # fixtures of the dependence / distance-zone families
    for k in ["cat2500_int", "cat2500_seg", "ibsi_fig3_19_ngldm_sample_image_int", "ibsi_fig3_19_ngldm_sample_image_mask",
              "nonibsi_rayryeng_ngldm_sample_image_int", "nonibsi_rayryeng_ngldm_sample_image_mask",
              "ibsi_fig3_17a_gldzm_sample_image_int", "ibsi_fig3_17a_gldzm_sample_image_mask"]:
        res["pixels"][k] = px[k]

    dsb = strip_comments(open(os.path.join(REF, "test_dsb2018_data.h")).read())
    rois = []
    for m in re.finditer(r"\{\s*(\d+)\s*,\s*(\d+)\s*,\s*\{(.*?)\}\s*\}", dsb, flags=re.S):
        w, h = int(m.group(1)), int(m.group(2))
        vals = [int(v) for v in re.findall(r"\d+", m.group(3))]
        assert len(vals) == w * h, (w, h, len(vals))
        rois.append({"w": w, "h": h, "pixels": vals})
    res["dsb2018"] = rois

    for f in ["test_2d_firstorder_matlab.h", "test_2d_firstorder_regression.h", "test_2d_firstorder_pyradiomics.h",
              "test_2d_glcm_regression.h", "test_2d_glcm_ibsi.h", "test_2d_glrlm_regression.h", "test_2d_glrlm_ibsi.h",
              "test_2d_glszm_regression.h", "test_2d_glszm_ibsi.h", "test_2d_ngtdm_regression.h", "test_2d_ngtdm_ibsi.h",
              "test_2d_gldm_regression.h", "test_2d_gldm_ibsi.h", "test_2d_ngldm_regression.h", "test_2d_ngldm_ibsi.h",
              "test_2d_ngldm_mirp.h", "test_2d_gldzm_ibsi.h"]:
        p = os.path.join(REF, f)
        if not os.path.exists(p):
            continue
        res["goldens"].update(scalar_maps(strip_comments(open(p).read())))
    # 2-D geometric moments: lists of {Feature2D code, "name", value} (test_2d_moments_{regression,skimage}.h)
    res["moment_goldens"] = {}
    for f in ["test_2d_moments_regression.h", "test_2d_moments_skimage.h"]:
        src = strip_comments(open(os.path.join(REF, f)).read())
        for m in re.finditer(r"ref_vals_list<GeomomentGoldenValue>\s+(\w+)\s*\{(.*?)\};", src, flags=re.S):
            res["moment_goldens"][m.group(1)] = {k: float(v) for k, v in re.findall(r"\{\s*Nyxus::Feature2D::\w+\s*,\s*\"(\w+)\"\s*,\s*([-+0-9.eE]+)\s*\}", m.group(2))}
    z = strip_comments(open(os.path.join(REF, "test_2d_zernike_regression.h")).read())
    res["vector_goldens"].update(vector_maps(z))
    g = strip_comments(open(os.path.join(REF, "test_gabor_truth.h")).read())
    m = re.search(r"gabor_truth\s*=\s*\{(.*?)\};", g, flags=re.S)
    res["gabor_truth"] = [[float(x) for x in re.findall(r"[-+0-9.eE]+", row)] for row in re.findall(r"\{([^{}]*)\}", m.group(1))]

    with open(OUT, "w") as fh:
        json.dump(res, fh, indent=0, separators=(",", ":"))
    print("wrote", OUT, {k: (len(v) if hasattr(v, "__len__") else v) for k, v in res.items()})
    for k, v in res["goldens"].items():
        print("  ", k, len(v))


if __name__ == "__main__":
    main()
