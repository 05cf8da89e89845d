"""Feature names, groups and column selection for the families the HIP path covers.

Names and order mirror the reference's `Feature2D` enum
(/root/reference/src/nyx/featureset.h:12-46, 174-233, 236-268, 271-288, 291-306, 309-343, 346-357), its user-facing
names (src/nyx/featureset.cpp `UserFacingFeatureNames`) and group tokens
(`UserFacing2dFeaturegroupNames`, featureset.cpp:650-665).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

from . import _abi

INTENSITY = ["COV", "COVERED_IMAGE_INTENSITY_RANGE", "ENERGY", "ENTROPY", "EXCESS_KURTOSIS", "HYPERFLATNESS",
             "HYPERSKEWNESS", "INTEGRATED_INTENSITY", "INTERQUARTILE_RANGE", "KURTOSIS", "MAX", "MEAN",
             "MEAN_ABSOLUTE_DEVIATION", "MEDIAN", "MEDIAN_ABSOLUTE_DEVIATION", "MIN", "MODE", "P01", "P10", "P25",
             "P75", "P90", "P99", "QCOD", "RANGE", "ROBUST_MEAN", "ROBUST_MEAN_ABSOLUTE_DEVIATION",
             "ROOT_MEAN_SQUARED", "SKEWNESS", "STANDARD_DEVIATION", "STANDARD_DEVIATION_BIASED", "STANDARD_ERROR",
             "VARIANCE", "VARIANCE_BIASED", "UNIFORMITY", "UNIFORMITY_PIU"]
GLCM_ANGLED = ["GLCM_ASM", "GLCM_ACOR", "GLCM_CLUPROM", "GLCM_CLUSHADE", "GLCM_CLUTEND", "GLCM_CONTRAST",
               "GLCM_CORRELATION", "GLCM_DIFAVE", "GLCM_DIFENTRO", "GLCM_DIFVAR", "GLCM_DIS", "GLCM_ENERGY",
               "GLCM_ENTROPY", "GLCM_HOM1", "GLCM_HOM2", "GLCM_ID", "GLCM_IDN", "GLCM_IDM", "GLCM_IDMN",
               "GLCM_INFOMEAS1", "GLCM_INFOMEAS2", "GLCM_IV", "GLCM_JAVE", "GLCM_JE", "GLCM_JMAX", "GLCM_JVAR",
               "GLCM_SUMAVERAGE", "GLCM_SUMENTROPY", "GLCM_SUMVARIANCE", "GLCM_VARIANCE"]
GLCM_AVE = ["GLCM_ASM_AVE", "GLCM_ACOR_AVE", "GLCM_CLUPROM_AVE", "GLCM_CLUSHADE_AVE", "GLCM_CLUTEND_AVE",
            "GLCM_CONTRAST_AVE", "GLCM_CORRELATION_AVE", "GLCM_DIFAVE_AVE", "GLCM_DIFENTRO_AVE", "GLCM_DIFVAR_AVE",
            "GLCM_DIS_AVE", "GLCM_ENERGY_AVE", "GLCM_ENTROPY_AVE", "GLCM_HOM1_AVE", "GLCM_ID_AVE", "GLCM_IDN_AVE",
            "GLCM_IDM_AVE", "GLCM_IDMN_AVE", "GLCM_IV_AVE", "GLCM_JAVE_AVE", "GLCM_JE_AVE", "GLCM_INFOMEAS1_AVE",
            "GLCM_INFOMEAS2_AVE", "GLCM_VARIANCE_AVE", "GLCM_JMAX_AVE", "GLCM_JVAR_AVE", "GLCM_SUMAVERAGE_AVE",
            "GLCM_SUMENTROPY_AVE", "GLCM_SUMVARIANCE_AVE"]
GLRLM_ANGLED = ["GLRLM_SRE", "GLRLM_LRE", "GLRLM_GLN", "GLRLM_GLNN", "GLRLM_RLN", "GLRLM_RLNN", "GLRLM_RP",
                "GLRLM_GLV", "GLRLM_RV", "GLRLM_RE", "GLRLM_LGLRE", "GLRLM_HGLRE", "GLRLM_SRLGLE", "GLRLM_SRHGLE",
                "GLRLM_LRLGLE", "GLRLM_LRHGLE"]
GLRLM_AVE = [n + "_AVE" for n in GLRLM_ANGLED]
GLSZM = ["GLSZM_SAE", "GLSZM_LAE", "GLSZM_GLN", "GLSZM_GLNN", "GLSZM_SZN", "GLSZM_SZNN", "GLSZM_ZP", "GLSZM_GLV",
         "GLSZM_ZV", "GLSZM_ZE", "GLSZM_LGLZE", "GLSZM_HGLZE", "GLSZM_SALGLE", "GLSZM_SAHGLE", "GLSZM_LALGLE",
         "GLSZM_LAHGLE"]
GLDZM = ["GLDZM_SDE", "GLDZM_LDE", "GLDZM_LGLZE", "GLDZM_HGLZE", "GLDZM_SDLGLE", "GLDZM_SDHGLE", "GLDZM_LDLGLE", "GLDZM_LDHGLE",
         "GLDZM_GLNU", "GLDZM_GLNUN", "GLDZM_ZDNU", "GLDZM_ZDNUN", "GLDZM_ZP", "GLDZM_GLM", "GLDZM_GLV", "GLDZM_ZDM", "GLDZM_ZDV",
         "GLDZM_ZDE"]
GLDM = ["GLDM_SDE", "GLDM_LDE", "GLDM_GLN", "GLDM_DN", "GLDM_DNN", "GLDM_GLV", "GLDM_DV", "GLDM_DE", "GLDM_LGLE", "GLDM_HGLE",
        "GLDM_SDLGLE", "GLDM_SDHGLE", "GLDM_LDLGLE", "GLDM_LDHGLE"]
NGLDM = ["NGLDM_LDE", "NGLDM_HDE", "NGLDM_LGLCE", "NGLDM_HGLCE", "NGLDM_LDLGLE", "NGLDM_LDHGLE", "NGLDM_HDLGLE", "NGLDM_HDHGLE",
         "NGLDM_GLNU", "NGLDM_GLNUN", "NGLDM_DCNU", "NGLDM_DCNUN", "NGLDM_DCP", "NGLDM_GLM", "NGLDM_GLV", "NGLDM_DCM", "NGLDM_DCV",
         "NGLDM_DCENT", "NGLDM_DCENE"]
NGTDM = ["NGTDM_COARSENESS", "NGTDM_CONTRAST", "NGTDM_BUSYNESS", "NGTDM_COMPLEXITY", "NGTDM_STRENGTH"]

_PQ13 = ["00", "01", "02", "03", "10", "11", "12", "13", "20", "21", "22", "23", "30"]
_PQ16 = ["%d%d" % (p, q) for p in range(4) for q in range(4)]
_PQ7 = ["02", "03", "11", "12", "20", "21", "30"]
_PQ10 = ["00", "01", "02", "03", "10", "11", "12", "20", "21", "30"]
SMOMS = (["SPAT_MOMENT_" + k for k in _PQ13] + ["CENTRAL_MOMENT_" + k for k in _PQ16] + ["NORM_SPAT_MOMENT_" + k for k in _PQ16]
         + ["NORM_CENTRAL_MOMENT_" + k for k in _PQ7] + ["HU_M%d" % k for k in range(1, 8)] + ["WEIGHTED_SPAT_MOMENT_" + k for k in _PQ10]
         + ["WEIGHTED_CENTRAL_MOMENT_" + k for k in _PQ7] + ["WT_NORM_CTR_MOM_" + k for k in _PQ7] + ["WEIGHTED_HU_M%d" % k for k in range(1, 8)])
IMOMS = (["IMOM_RM_" + k for k in _PQ13] + ["IMOM_CM_" + k for k in _PQ16] + ["IMOM_NRM_" + k for k in _PQ16] + ["IMOM_NCM_" + k for k in _PQ7]
         + ["IMOM_HU%d" % k for k in range(1, 8)] + ["IMOM_WRM_" + k for k in _PQ10] + ["IMOM_WCM_" + k for k in _PQ7]
         + ["IMOM_WNCM_" + k for k in _PQ7] + ["IMOM_WHU%d" % k for k in range(1, 8)])

# feature name -> family bit
FAMILY_OF: Dict[str, int] = {}
for _n in INTENSITY:
    FAMILY_OF[_n] = _abi.FAM_INTENSITY
for _n in GLCM_ANGLED + GLCM_AVE:
    FAMILY_OF[_n] = _abi.FAM_GLCM
for _n in GLRLM_ANGLED + GLRLM_AVE:
    FAMILY_OF[_n] = _abi.FAM_GLRLM
for _n in GLSZM:
    FAMILY_OF[_n] = _abi.FAM_GLSZM
for _n in NGTDM:
    FAMILY_OF[_n] = _abi.FAM_NGTDM
for _n in GLDZM:
    FAMILY_OF[_n] = _abi.FAM_GLDZM
for _n in GLDM:
    FAMILY_OF[_n] = _abi.FAM_GLDM
for _n in NGLDM:
    FAMILY_OF[_n] = _abi.FAM_NGLDM
for _n in SMOMS:
    FAMILY_OF[_n] = _abi.FAM_SMOMS
for _n in IMOMS:
    FAMILY_OF[_n] = _abi.FAM_IMOMS
FAMILY_OF["GABOR"] = _abi.FAM_GABOR
FAMILY_OF["ZERNIKE2D"] = _abi.FAM_ZERNIKE

# group tokens (featureset.cpp:650-665) the HIP path can serve completely
GROUPS: Dict[str, List[str]] = {
    "*ALL_INTENSITY*": INTENSITY,
    "*ALL_GLCM*": GLCM_ANGLED + GLCM_AVE,
    "*ALL_GLRLM*": GLRLM_ANGLED + GLRLM_AVE,
    "*ALL_GLDZM*": GLDZM,
    "*ALL_GLSZM*": GLSZM,
    "*ALL_GLDM*": GLDM,
    "*ALL_NGLDM*": NGLDM,
    "*ALL_NGTDM*": NGTDM,
    "*GEOMOMS*": SMOMS + IMOMS,      # env_features.cpp:316-333
    "*SGEOMOMS*": SMOMS,
    "*IGEOMOMS*": IMOMS,
}

# enum order of every feature code the path covers (one entry per Feature2D code)
ENUM_ORDER: List[str] = (INTENSITY + GLCM_ANGLED + GLCM_AVE + GLRLM_ANGLED + GLRLM_AVE + GLDZM + GLSZM + GLDM + NGLDM + NGTDM
                         + ["GABOR", "ZERNIKE2D"] + SMOMS + IMOMS)


def expand(features: List[str]) -> Tuple[int, List[str]]:
    """Expands group tokens, validates names, returns (family mask, requested feature codes in enum order)."""
    want = set()
    unknown = []
    for f in features:
        key = f.strip()
        if key.upper() in GROUPS:
            want.update(GROUPS[key.upper()])
        elif key.upper() in FAMILY_OF:
            want.add(key.upper())
        else:
            unknown.append(f)
    if unknown:
        raise ValueError(
            f"feature(s) {unknown} are not served by the MI355X path. Implemented: groups {sorted(GROUPS)} and the "
            f"individual features of the intensity, GLCM, GLRLM, GLDZM, GLSZM, GLDM, NGLDM and NGTDM families, GABOR, ZERNIKE2D")
    if not want:
        raise ValueError("no features requested")
    ordered = [n for n in ENUM_ORDER if n in want]
    mask = 0
    for n in ordered:
        mask |= FAMILY_OF[n]
    return mask, ordered


def column_selector(requested: List[str], all_columns: List[str], glcm_angles: List[int]) -> List[int]:
    """Indices into the library's table (all columns of the touched families) of the columns that belong
    to the requested feature codes, in output order (save_features_2_buffer, output_2_buffer.cpp:316-445)."""
    idx_of = {c: i for i, c in enumerate(all_columns)}
    sel: List[int] = []
    for code in requested:
        if code in GLCM_ANGLED:
            sel += [idx_of[f"{code}_{a}"] for a in glcm_angles]
        elif code in GLRLM_ANGLED:
            sel += [idx_of[f"{code}_{a}"] for a in (0, 45, 90, 135)]
        elif code == "GABOR":
            sel += [i for i, c in enumerate(all_columns) if c.startswith("GABOR_")]
        elif code == "ZERNIKE2D":
            sel += [i for i, c in enumerate(all_columns) if c.startswith("ZERNIKE2D_Z")]
        else:
            sel.append(idx_of[code])
    return sel
