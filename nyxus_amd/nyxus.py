"""`Nyxus` -- the reference's 2-D Python class, served by the MI355X hot path.

Mirrors `class Nyxus` of /root/reference/src/nyx/python/nyxus/nyxus.py: constructor keyword whitelist and
checks (:162-246), `featurize` (:385-519: argument validation, 2-D -> 3-D promotion, default names, the
negative-intensity shift and uint32 casts, DataFrame layout), `featurize_directory` (:303-382).
The feature reduce itself is `nyxhip_featurize_batch` (include/nyxhip.h); ROI assembly is
`roi_assembly.assemble`.  Features outside the hot-path families raise ValueError -- there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import re
import sys
from typing import List, Optional

import numpy as np

from . import _abi, _lib, featureset, roi_assembly

_VALID_KEYS = {
    "neighbor_distance", "pixels_per_micron", "coarse_gray_depth", "n_feature_calc_threads", "use_gpu_device", "ibsi",
    "gabor_kersize", "gabor_gamma", "gabor_sig2lam", "gabor_f0", "gabor_thold", "gabor_thetas", "gabor_freqs",
    "channel_signature", "parent_channel", "child_channel", "aggregate", "dynamic_range", "min_intensity",
    "max_intensity", "ram_limit", "verbose", "anisotropy_x", "anisotropy_y", "mergerois", "preserve_hu"}

_DBL_MAX = sys.float_info.max


def _f32(x) -> float:
    """The reference parses Gabor parameters with strtof (helpers.h:82-99): float32 precision."""
    return float(np.float32(x))


class Nyxus:
    def __init__(self, features: List[str], **kwargs):
        invalid = set(kwargs) - _VALID_KEYS
        if invalid:
            print(f"Warning: unexpected keyword argument(s): {', '.join(invalid)}")
        neighbor_distance = kwargs.get("neighbor_distance", 5)
        pixels_per_micron = kwargs.get("pixels_per_micron", 1.0)
        coarse_gray_depth = kwargs.get("coarse_gray_depth", 64)
        n_threads = kwargs.get("n_feature_calc_threads", 4)
        use_gpu_device = kwargs.get("use_gpu_device", -1)
        verb = kwargs.get("verbose", 0)
        if neighbor_distance <= 0:
            raise ValueError("Neighbor distance must be greater than zero.")
        if pixels_per_micron <= 0:
            raise ValueError("Pixels per micron must be greater than zero.")
        if coarse_gray_depth <= 0:
            raise ValueError("Custom number of grayscale levels (parameter coarse_gray_depth, default=64) must be non-negative.")
        if n_threads < 1:
            raise ValueError("There must be at least one feature calculation thread.")
        if verb < 0:
            raise ValueError("verbosity must be non-negative")
        if kwargs.get("anisotropy_x", 1.0) <= 0:
            raise ValueError("anisotropy_x must be positive")
        if kwargs.get("anisotropy_y", 1.0) <= 0:
            raise ValueError("anisotropy_y must be positive")
        if kwargs.get("anisotropy_x", 1.0) != 1.0 or kwargs.get("anisotropy_y", 1.0) != 1.0:
            raise ValueError("anisotropy is outside the MI355X hot path (SURVEY.md section 8)")

        self._features = list(features)
        self._mask, self._requested = featureset.expand(self._features)
        self._settings = _abi.default_settings(int(coarse_gray_depth), bool(kwargs.get("ibsi", False)))
        self._env = {"neighbor_distance": neighbor_distance, "pixels_per_micron": pixels_per_micron, "n_feature_calc_threads": n_threads,
                     "dynamic_range": kwargs.get("dynamic_range", 10000), "min_intensity": kwargs.get("min_intensity", 0.0),
                     "max_intensity": kwargs.get("max_intensity", 1.0), "ram_limit": kwargs.get("ram_limit", -1), "verbose": verb}
        self.set_gabor_feature_params(
            kersize=kwargs.get("gabor_kersize", 16), gamma=kwargs.get("gabor_gamma", 0.1),
            sig2lam=kwargs.get("gabor_sig2lam", 0.8), f0=kwargs.get("gabor_f0", 0.1),
            thold=kwargs.get("gabor_thold", 0.025), thetas=kwargs.get("gabor_thetas", [0, 45, 90, 135]),
            freqs=kwargs.get("gabor_freqs", [4, 16, 32, 64]))
        self._device = max(int(use_gpu_device), 0)   # the GPU is the only compute path of this package
        self._ctx: Optional[_lib.Context] = None
        self._valid_output_types = ["pandas", "arrowipc", "parquet"]
        self.error_message = ""

    # -- Gabor bank: customize_gabor_feature_imp -> parse_gabor_options_raw_inputs (cli_gabor_options.cpp:14-120)
    def set_gabor_feature_params(self, **kw):
        params = ["kersize", "gamma", "sig2lam", "f0", "thold", "thetas", "freqs"]
        for k in kw:
            if k not in params:
                raise ValueError(f"Invalid Gabor parameter {k}. The valid parameters are: {params}")
        s = self._settings
        if "kersize" in kw:
            s.gabor_kersize = int(kw["kersize"])
        if "gamma" in kw:
            s.gabor_gamma = _f32(kw["gamma"])
        if "sig2lam" in kw:
            s.gabor_sig2lam = _f32(kw["sig2lam"])
        if "f0" in kw:
            s.gabor_f0lp = _f32(kw["f0"])
        if "thold" in kw:
            s.gabor_graythr = _f32(kw["thold"])
        if ("thetas" in kw) != ("freqs" in kw):
            raise ValueError("Invalid GABOR parameter value: frequency and angle lists are allowed to be both empty or non-empty")
        if "thetas" in kw:
            th, fr = list(kw["thetas"]), list(kw["freqs"])
            if len(th) != len(fr):
                raise ValueError(f"Invalid GABOR parameter value: frequency and angle lists must me of same size. Received thetas={th} freqs={fr}")
            if len(th) > _abi.MAX_GABOR_FILTERS:
                raise ValueError(f"at most {_abi.MAX_GABOR_FILTERS} Gabor filters are supported")
            s.gabor_n_filters = len(th)
            for i, (t, f) in enumerate(zip(th, fr)):
                s.gabor_f0[i] = _f32(f)
                s.gabor_theta[i] = _f32(t) / 180.0 * 3.14159265358979323846   # deg2rad, helpers.h:371-374

    def _context(self) -> _lib.Context:
        if self._ctx is None:
            self._ctx = _lib.Context(self._device)    # raises without a GPU / library: no CPU fallback
        return self._ctx

    def _columns(self):
        names = _lib.column_names(self._mask, self._settings)
        angles = [self._settings.glcm_angles[i] for i in range(self._settings.glcm_n_angles)]
        sel = featureset.column_selector(self._requested, names, angles)
        return [names[i] for i in sel], sel

    def _featurize_pair(self, inten: np.ndarray, label: np.ndarray, slide_min, slide_max):
        batch = roi_assembly.assemble(inten, label, slide_min, slide_max)
        if batch is None:
            return np.zeros((0,), np.uint32), np.zeros((0, 0))
        table = self._context().featurize_host(batch, self._mask, self._settings)
        # NaN / inf -> noval (force_finite_number, helpers.h:376-382; save_features_2_buffer applies it per value)
        _lib.load().nyxhip_finalize_table(table.ctypes.data, table.shape[0], table.shape[1], table.shape[1],
                                          C.c_double(self._settings.soft_nan))
        return batch.roi_label, table

    def featurize(self, intensity_images: np.ndarray, label_images: np.ndarray, intensity_names: list = [],
                  label_names: list = [], output_type: Optional[str] = "pandas", output_path: Optional[str] = ""):
        import pandas as pd
        if output_type != "" and output_type not in self._valid_output_types:
            raise ValueError(f"Invalid output type: {output_type}. Valid options are: {self._valid_output_types}")
        if not isinstance(intensity_images, np.ndarray):
            raise ValueError("intensity_images parameter must be numpy.ndarray")
        if not isinstance(label_images, np.ndarray):
            raise ValueError("label_images parameter must be numpy.ndarray")
        if output_type not in self._valid_output_types:
            raise ValueError(f"Invalid output type {output_type}. Valid output types are {self._valid_output_types}.")
        if output_type != "pandas":
            raise ValueError("arrowipc / parquet writers are outside the MI355X hot path (SURVEY.md section 8); use 'pandas'")
        if intensity_images.ndim == 2:
            if label_images.ndim != 2:
                raise ValueError("Both intensity and label arrays must be the same dimension")
            intensity_images = np.array([intensity_images])
            label_images = np.array([label_images])
        elif intensity_images.ndim == 3:
            if label_images.ndim != 3:
                raise ValueError("Both intensity and label arrays must be the same dimension")
        else:
            raise ValueError("Intensity and label arrays must be 2D or 3D")
        if intensity_images.shape != label_images.shape:
            raise ValueError("Intensity and label image arrays must have the same number of images with matching dimensions")
        if intensity_names == []:
            intensity_names = ["Intensity" + str(i) for i in range(intensity_images.shape[0])]
        if label_names == []:
            label_names = ["Segmentation" + str(i) for i in range(label_images.shape[0])]
        if intensity_images.shape[0] != len(intensity_names):
            raise ValueError("Number of _intensity image names_ (" + str(len(intensity_names)) + ") must be the same as the number of intensity images (" + str(intensity_images.shape[0]) + ")")
        if label_images.shape[0] != len(label_names):
            raise ValueError("Number of segmentation names must be the same as the number of images.")
        # Hounsfield-style input: shift to non-negative, then the unsigned casts (nyxus.py:480-489)
        # (an unsigned array has no negative minimum to look for, and uint32 input is handed over without a copy: at the
        # device rates of this path every avoidable host pass over the images shows)
        I = intensity_images
        if not np.issubdtype(I.dtype, np.unsignedinteger):
            min_raw = np.min(I)
            if min_raw < 0:
                I = I - min_raw
        if I.dtype != np.uint32:
            I = I.astype(np.uint32)
        M = label_images if label_images.dtype == np.uint32 else label_images.astype(np.uint32)

        cols, sel = self._columns()
        # All images of the stack go through the fused device path in one call: label scan, ROI assembly and the
        # reduce run on the GPU (nyxhip_featurize_tiles).  The montage prescan of the reference leaves slide
        # min/max at +/-DBL_MAX (slideprops.cpp:27-28,74-75), so COVERED_IMAGE_INTENSITY_RANGE = range / -inf = -0.0
        # in this entry point; the tile ABI implements exactly that.
        tiles, labels, table = self._context().featurize_tiles_host(I, M, self._mask, self._settings)
        _lib.load().nyxhip_finalize_table(table.ctypes.data, table.shape[0], table.shape[1], table.shape[1],
                                          C.c_double(self._settings.soft_nan))
        # [intensity_image, mask_image, ROI_label (uint32), t_index, features...]: the feature block is wrapped as it is, the
        # four leading columns are inserted in front of it (no per-row Python work, no float round trip of the labels)
        ti = np.asarray(tiles, dtype=np.intp)
        df = pd.DataFrame(table[:, sel], columns=cols)
        df.insert(0, "t_index", np.zeros(len(labels)))
        df.insert(0, "ROI_label", np.asarray(labels, dtype=np.uint32))
        df.insert(0, "mask_image", np.asarray(label_names, dtype=object)[ti] if len(ti) else np.empty(0, dtype=object))
        df.insert(0, "intensity_image", np.asarray(intensity_names, dtype=object)[ti] if len(ti) else np.empty(0, dtype=object))
        return df

    def featurize_directory(self, intensity_dir: str, label_dir: Optional[str] = None, file_pattern: Optional[str] = ".*",
                            output_type: Optional[str] = "pandas", output_path: Optional[str] = ""):
        import pandas as pd
        if not os.path.exists(intensity_dir):
            raise IOError(f"Provided intensity image directory '{intensity_dir}' does not exist.")
        if label_dir is not None and not os.path.exists(label_dir):
            raise IOError(f"Provided label image directory '{label_dir}' does not exist.")
        if label_dir is None:
            label_dir = intensity_dir
        if output_type not in self._valid_output_types:
            raise ValueError(f"Invalid output type {output_type}. Valid output types are {self._valid_output_types}.")
        if output_type != "pandas":
            raise ValueError("arrowipc / parquet writers are outside the MI355X hot path (SURVEY.md section 8); use 'pandas'")
        rx = re.compile(file_pattern)
        files = sorted(f for f in os.listdir(intensity_dir) if rx.fullmatch(f) and os.path.isfile(os.path.join(label_dir, f)))
        return self._featurize_file_pairs([os.path.join(intensity_dir, f) for f in files], [os.path.join(label_dir, f) for f in files])

    def _featurize_file_pairs(self, intensity_files: list, mask_files: list):
        import pandas as pd
        from . import tiff_ingest
        cols, sel = self._columns()
        str_rows, num_rows = [], []
        for fi, fm in zip(intensity_files, mask_files):
            I = tiff_ingest.read_tiff(fi)
            M = tiff_ingest.read_tiff(fm).astype(np.uint32)
            if I.shape != M.shape:
                raise ValueError(f"{fi}: intensity and mask images differ in shape")
            # slide prescan: min/max of the intensities under any mask (scan_slide_props, slideprops.cpp:456-...)
            fg = I[M != 0]
            smin, smax = (float(fg.min()), float(fg.max())) if fg.size else (0.0, 0.0)
            labels, table = self._featurize_pair(I.astype(np.uint32), M, smin, smax)
            for r in range(len(labels)):
                str_rows.append([os.path.basename(fi), os.path.basename(fm)])
                num_rows.append(np.concatenate(([float(labels[r]), 0.0], table[r, sel])))
        header = ["intensity_image", "mask_image", "ROI_label", "t_index"] + cols
        string_data = np.array(str_rows, dtype=object).reshape(-1, 2)
        numeric_data = np.array(num_rows, dtype=np.float64).reshape(-1, 2 + len(cols))
        df = pd.concat([pd.DataFrame(string_data, columns=header[:2]), pd.DataFrame(numeric_data, columns=header[2:])], axis=1)
        if "ROI_label" in df.columns:
            df.ROI_label = df.ROI_label.astype(np.uint32)
        return df

    def featurize_files(self, intensity_files: list, mask_files: list, single_roi: bool, output_type: Optional[str] = "pandas",
                        output_path: Optional[str] = ""):
        """Image file pairs passed as lists (reference nyxus.py:524-593)."""
        if intensity_files is None:
            raise IOError("The list of intensity file paths is empty")
        if mask_files is None and not single_roi:
            raise IOError("The list of segment file paths is empty. Supply mask images or set single_roi to True")
        if output_type not in self._valid_output_types:
            raise ValueError(f"Invalid output type {output_type}. Valid output types are {self._valid_output_types}")
        if output_type != "pandas":
            raise ValueError("arrowipc / parquet writers are outside the MI355X hot path (SURVEY.md section 8); use 'pandas'")
        if single_roi:
            raise ValueError("single-ROI (whole-slide) featurization is outside the MI355X hot path (SURVEY.md section 8)")
        if len(intensity_files) != len(mask_files):
            raise ValueError("intensity_files and mask_files must have the same length")
        return self._featurize_file_pairs(list(intensity_files), list(mask_files))

    # -- parameters (reference nyxus.py:264-300, :521-522, :703-868) -------------------------------------------------------
    def use_gpu_device(self, gpu_device_id: int):
        self._device = max(int(gpu_device_id), 0)
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None

    def set_metaparam(self, paramval: str):
        """Feature-specific parameter, e.g. "glcm/greydepth=25" (env_metaparams.cpp:63-246)."""
        err = None
        sides = paramval.split("=")
        if len(sides) != 2:
            err = f'syntax error in "{paramval}": expecting <paramName>=<paramVal>'
        else:
            path = sides[0].split("/")
            if len(path) == 2 and path[0] == "glcm" and path[1] in ("greydepth", "offset"):
                try:
                    v = int(sides[1])
                except ValueError:
                    err = f'error: cannot parse value "{sides[1]}" of glcm/{path[1]}: expecting an integer'
                else:
                    if path[1] == "greydepth":
                        self._settings.glcm_grey_depth = v     # the degenerate-ROI guard's depth (glcm.cpp:23)
                    else:
                        self._settings.glcm_offset = v
            elif len(path) == 2 and path[0] == "glcm":
                err = f'error: unrecognized feature parameter of feature glcm: "{path[1]}"'
            elif len(path) == 2:
                err = f'error: unrecognized feature "{path[0]}"' if not path[0].startswith("3") else "3-D features are outside the MI355X hot path"
            else:
                err = f'syntax error in <paramName>=<paramVal> of "{paramval}": expecting <paramName> to be <feature name>/<parameter name>'
        if err:
            raise ValueError(f"Invalid metaparameter value {paramval}: {err}")

    def get_metaparam(self, paramname: str):
        path = paramname.split("/")
        if len(path) == 2 and path[0] == "glcm" and path[1] == "greydepth":
            return float(self._settings.glcm_grey_depth)
        if len(path) == 2 and path[0] == "glcm" and path[1] == "offset":
            return float(self._settings.glcm_offset)
        raise NameError(f"Invalid metaparameter name {paramname}: error: unrecognized feature parameter")

    def set_environment_params(self, **params):
        valid_params = ["features", "neighbor_distance", "pixels_per_micron", "coarse_gray_depth", "n_feature_calc_threads", "use_gpu_device",
                        "verbose", "dynamic_range", "min_intensity", "max_intensity", "ram_limit"]
        for key in params:
            if key not in valid_params:
                raise ValueError(f"Invalid environment parameter {key}. Value parameters are {params}")
        if params.get("features"):
            self._features = list(params["features"])
            self._mask, self._requested = featureset.expand(self._features)
        if params.get("coarse_gray_depth", 0) != 0:      # 0 = "leave as is" (new_bindings_py.cpp set_environment_params_imp)
            gd = int(params["coarse_gray_depth"])
            self._settings.grey_depth = gd
            self._settings.glcm_grey_depth = gd
        if params.get("use_gpu_device", -1) >= 0:
            self.use_gpu_device(params["use_gpu_device"])
        for k, absent in (("neighbor_distance", -1), ("pixels_per_micron", -1), ("n_feature_calc_threads", 0), ("dynamic_range", -1),
                          ("min_intensity", -1), ("max_intensity", -1), ("ram_limit", -1)):
            if k in params and params[k] != absent:      # the reference passes these sentinels for "not given" (nyxus.py:745-755)
                self._env[k] = params[k]
        if "verbose" in params:
            self._env["verbose"] = params["verbose"]

    def set_params(self, **params):
        available = ["features", "neighbor_distance", "pixels_per_micron", "coarse_gray_depth", "n_feature_calc_threads", "use_gpu_device",
                     "ibsi", "dynamic_range", "min_intensity", "max_intensity", "ram_limit", "verbose"]
        env, gab = {}, {}
        for key, value in params.items():
            if key.startswith("gabor_"):
                gab[key[len("gabor_"):]] = value
            elif key == "ibsi":
                self._settings.ibsi = 1 if value else 0
            elif key not in available:
                raise ValueError("Invalid parameter: ", key)
            else:
                env[key] = value
        if gab:
            self.set_gabor_feature_params(**gab)
        if env:
            self.set_environment_params(**env)

    def get_params(self, *args):
        s = self._settings
        params = {"features": list(self._features), "neighbor_distance": self._env["neighbor_distance"],
                  "pixels_per_micron": self._env["pixels_per_micron"], "coarse_gray_depth": int(s.grey_depth),
                  "n_feature_calc_threads": self._env["n_feature_calc_threads"], "ibsi": bool(s.ibsi),
                  "gabor_kersize": int(s.gabor_kersize), "gabor_gamma": float(s.gabor_gamma), "gabor_sig2lam": float(s.gabor_sig2lam),
                  "gabor_f0": float(s.gabor_f0lp), "gabor_thold": float(s.gabor_graythr),
                  "gabor_freqs": [float(s.gabor_f0[i]) for i in range(s.gabor_n_filters)],
                  "gabor_thetas": [float(s.gabor_theta[i]) * 180.0 / 3.14159265358979323846 for i in range(s.gabor_n_filters)],   # rad2deg
                  "dynamic_range": self._env["dynamic_range"], "min_intensity": self._env["min_intensity"],
                  "max_intensity": self._env["max_intensity"], "ram_limit": self._env["ram_limit"],
                  "using_gpu": True, "gpu_device_id": self._device}
        if not args:
            return params
        return {k: params[k] for k in args if k in params}
