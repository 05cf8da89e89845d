"""The C++ plugin adapter (include/nyxhip_feature_method.hpp): compiles against the ABI on CPU; on the GPU
box the reference-style C++ unit tests (tests/cpp/test_feature_method.cpp) run against the golden tables."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
BIN = os.path.join(CPP, "test_feature_method.bin")


def _gen_fixture():
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_tests.json")))
    out = []
    for name in ["pixelIntensityFeaturesTestData"] + [f"ibsi_phantom_z{z}_{k}" for z in range(1, 5) for k in ("intensity", "mask")]:
        rows = ", ".join("{%d, %d, %d}" % tuple(t) for t in ref["pixels"][name])
        out.append(f"static const NyxusPixel {name}[] = {{{rows}}};")
    for g in ("firstorder_2d_matlab_ref_vals", "glcm_2d_regression_ref_vals", "gldm_2d_ibsi_ref_vals", "ngldm_2d_ibsi_ref_vals",
              "ngldm_2d_mirp_ref_vals", "ngldm_2d_regression_ref_vals", "gldzm_2d_ibsi_ref_vals"):
        rows = ", ".join('{Feature2D::%s, "%s", %r}' % (k, k, v) for k, v in ref["goldens"][g].items())
        out.append(f"static const std::vector<Golden> {g} = {{{rows}}};")
    rows = ", ".join("{%d, %d, {%s}}" % (d["w"], d["h"], ", ".join(map(str, d["pixels"]))) for d in ref["dsb2018"])
    out.append(f"static const std::vector<ImageData> dsb_data = {{{rows}}};")
    rows = ", ".join("{%s}" % ", ".join(repr(v) for v in r) for r in ref["gabor_truth"])
    out.append(f"static const std::vector<std::vector<double>> gabor_truth = {{{rows}}};")
    with open(os.path.join(CPP, "fixture.inc"), "w") as fh:
        fh.write("// generated from tests/golden/reference_tests.json by tests/test_cpp_adapter.py\n" + "\n".join(out) + "\n")


def _build():
    _gen_fixture()
    lib = os.path.join(ROOT, "nyxus_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I", CPP, os.path.join(CPP, "test_feature_method.cpp"),
           "-o", BIN, "-L", lib, "-lnyxhip", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)


def test_adapter_compiles_and_links_against_the_abi():
    _build()
    r = subprocess.run([BIN, "--compile-check"], capture_output=True, text=True)
    assert r.returncode == 0 and "compiled" in r.stdout


@pytest.mark.gpu
def test_reference_style_cpp_tests_pass_on_gpu():
    if not os.path.exists(BIN):
        _build()
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ALL PASSED" in r.stdout, r.stdout + r.stderr
