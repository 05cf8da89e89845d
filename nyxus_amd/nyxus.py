"""`Nyxus` -- the reference's 2-D Python class, served by the MI355X hot path.

Mirrors `class Nyxus` of /root/reference/src/nyx/python/nyxus/nyxus.py: constructor keyword whitelist and
checks (:162-246), `featurize` (:385-519: argument validation, 2-D -> 3-D promotion, default names, the
negative-intensity shift and uint32 casts, DataFrame layout), `featurize_directory` (:303-382).
Label scan, ROI assembly and the feature reduce all run on the device (`nyxhip_featurize_tiles_v2`, include/nyxhip.h).  Features outside the hot-path families raise ValueError -- there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import re
import sys
from typing import List, Optional

import numpy as np

from . import _abi, _lib, featureset

_VALID_KEYS = {
    "neighbor_distance", "pixels_per_micron", "coarse_gray_depth", "n_feature_calc_threads", "use_gpu_device", "ibsi",
    "gabor_kersize", "gabor_gamma", "gabor_sig2lam", "gabor_f0", "gabor_thold", "gabor_thetas", "gabor_freqs",
    "channel_signature", "parent_channel", "child_channel", "aggregate", "dynamic_range", "min_intensity",
    "max_intensity", "ram_limit", "verbose", "anisotropy_x", "anisotropy_y", "mergerois", "preserve_hu",
    "gpu_devices"}     # gpu_devices: this package's one addition -- several GPUs of the node share a featurize() call

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
        # Multi-GPU: `gpu_devices=[0, 1, ...]` block-partitions the image stack of a featurize() call over one context per listed
        # device (nyxhip_featurize_tiles_sharded; ROIs are independent -- the reference slices its label vector the same way,
        # parallel.h:34-41); rows come back in (image, label) order whatever the device count.
        self._devices = [int(d) for d in kwargs.get("gpu_devices", [])] or None
        if self._devices:
            self._device = self._devices[0]
        self._ctx: Optional[_lib.Context] = None
        self._extra_ctx: List[_lib.Context] = []
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
            self._extra_ctx = [_lib.Context(d) for d in (self._devices or [])[1:]]
        return self._ctx

    def _device_budget(self) -> int:
        """`ram_limit` (megabytes, reference nyxus.py:148; the reference batches ROIs by it, phase2_2d.cpp:694-705) bounds the
        device workspace of a call; unset (-1) lets the library take half of the free device memory."""
        rl = self._env.get("ram_limit", -1)
        return int(rl) << 20 if rl is not None and rl > 0 else 0

    def _featurize_stack(self, I: np.ndarray, M: np.ndarray, slide_mode: int):
        ctx = self._context()
        tiles, labels, table = ctx.featurize_tiles_host(I, M, self._mask, self._settings, slide_mode=slide_mode,
                                                        max_device_bytes=self._device_budget(), contexts=self._extra_ctx)
        _lib.load().nyxhip_finalize_table(table.ctypes.data, table.shape[0], table.shape[1], table.shape[1],
                                          C.c_double(self._settings.soft_nan))
        return tiles, labels, table

    def _columns(self):
        names = _lib.column_names(self._mask, self._settings)
        angles = [self._settings.glcm_angles[i] for i in range(self._settings.glcm_n_angles)]
        sel = featureset.column_selector(self._requested, names, angles)
        return [names[i] for i in sel], sel

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
        # uint8 / uint16 / uint32 images are handed over as they are -- the kernels widen, H2D carries the image's own bytes
        I = intensity_images
        if not np.issubdtype(I.dtype, np.unsignedinteger):
            min_raw = np.min(I)
            if min_raw < 0:
                I = I - min_raw
        if I.dtype not in (np.uint8, np.uint16, np.uint32):
            I = I.astype(np.uint32)
        M = label_images if label_images.dtype in (np.uint8, np.uint16, np.uint32) else label_images.astype(np.uint32)

        cols, sel = self._columns()
        # All images of the stack go through the fused device path in one call: label scan, ROI assembly and the
        # reduce run on the GPU (nyxhip_featurize_tiles_v2), in chunks that fit the device budget.  The montage prescan of
        # the reference leaves slide min/max at +/-DBL_MAX (slideprops.cpp:27-28,74-75), so COVERED_IMAGE_INTENSITY_RANGE
        # = range / -inf = -0.0 in this entry point: NYXHIP_SLIDE_MONTAGE.
        tiles, labels, table = self._featurize_stack(I, M, _abi.SLIDE_MONTAGE)
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
        """File pairs through the device front end: every slide is one tile of a NYXHIP_SLIDE_PER_TILE call, i.e. the label scan,
        the slide prescan (min / max of the intensities under any mask: scan_slide_props, slideprops.cpp:456-...), ROI assembly
        and the reduce all run on the GPU; consecutive slides of equal shape and element type share a call."""
        import pandas as pd
        from . import tiff_ingest
        cols, sel = self._columns()
        blocks, names_i, names_m, lab_all = [], [], [], []

        def flush(stack_i, stack_m, fis, fms):
            if not stack_i:
                return
            tiles, labels, table = self._featurize_stack(np.stack(stack_i), np.stack(stack_m), _abi.SLIDE_PER_TILE)
            ti = np.asarray(tiles, dtype=np.intp)
            blocks.append(table[:, sel])
            lab_all.append(np.asarray(labels, dtype=np.uint32))
            names_i.append(np.asarray([os.path.basename(f) for f in fis], dtype=object)[ti])
            names_m.append(np.asarray([os.path.basename(f) for f in fms], dtype=object)[ti])

        si, sm, fis, fms = [], [], [], []
        for fi, fm in zip(intensity_files, mask_files):
            I = tiff_ingest.read_tiff(fi)
            M = tiff_ingest.read_tiff(fm)
            if I.shape != M.shape:
                raise ValueError(f"{fi}: intensity and mask images differ in shape")
            if I.dtype not in (np.uint8, np.uint16, np.uint32):
                I = I.astype(np.uint32)
            if M.dtype not in (np.uint8, np.uint16, np.uint32):
                M = M.astype(np.uint32)
            if si and (I.shape != si[0].shape or I.dtype != si[0].dtype or M.dtype != sm[0].dtype or len(si) * I.nbytes > (1 << 30)):
                flush(si, sm, fis, fms)
                si, sm, fis, fms = [], [], [], []
            si.append(I); sm.append(M); fis.append(fi); fms.append(fm)
        flush(si, sm, fis, fms)
        n = sum(len(l) for l in lab_all)
        df = pd.DataFrame(np.concatenate(blocks) if blocks else np.zeros((0, len(cols))), columns=cols)
        df.insert(0, "t_index", np.zeros(n))
        df.insert(0, "ROI_label", np.concatenate(lab_all) if lab_all else np.zeros(0, np.uint32))
        df.insert(0, "mask_image", np.concatenate(names_m) if names_m else np.empty(0, dtype=object))
        df.insert(0, "intensity_image", np.concatenate(names_i) if names_i else np.empty(0, dtype=object))
        return df

    @staticmethod
    def to_csv(df, path: str) -> None:
        """Writes a feature DataFrame in the format of the reference's CSV writer (save_features_2_csv,
        /root/reference/src/nyx/output_2_csv.cpp:420-755): a header of double-quoted column names (:470-484); per ROI the two
        file names double-quoted (std::filesystem::path streams quoted, :563), ROI label and time index as integers (:573) and
        every feature value through printf("%g") (:430, :595) -- NaN / inf were already replaced by the table writer."""
        cols = list(df.columns)
        str_cols = [c for c in ("intensity_image", "mask_image") if c in cols]
        int_cols = [c for c in ("ROI_label", "t_index") if c in cols]
        val_cols = [c for c in cols if c not in str_cols and c not in int_cols]
        with open(path, "w", buffering=32768) as fh:
            fh.write(",".join('"%s"' % c for c in cols) + "\n")
            S = df[str_cols].values
            Iv = df[int_cols].values.astype(np.int64)
            V = df[val_cols].values.astype(np.float64)
            for r in range(len(df)):
                parts = ['"%s"' % v for v in S[r]] + ["%d" % v for v in Iv[r]] + ["%g" % v for v in V[r]]
                fh.write(",".join(parts) + "\n")

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
