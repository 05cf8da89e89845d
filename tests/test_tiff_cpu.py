"""The native TIFF tile ingest (include/nyxtiff.h, nyxus_amd/libnyxtiff.so) on the CPU: the reference's own OME-TIFF fixtures,
tiled and stripped files of every unsigned width, signed clamping, the float refusal; the library exports what the header declares."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

from nyxus_amd import tiff_ingest
from tests import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "nyxtiff.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(nyxtiff_[a-z_0-9]+)\s*\(", hdr)))
    assert declared == ["nyxtiff_info", "nyxtiff_read"]
    lib = C.CDLL(os.path.join(ROOT, "nyxus_amd", "libnyxtiff.so"))
    for name in declared:
        assert hasattr(lib, name)


def test_reference_fixtures_decode_like_pillow():
    from PIL import Image
    for sub in ("int", "seg"):
        d = os.path.join(ROOT, "tests", "golden", "tiff", sub)
        for f in sorted(os.listdir(d)):
            a = tiff_ingest.read_tiff(os.path.join(d, f))
            assert a.dtype == np.uint16 and a.shape == (256, 256)         # a 256 x 256 image inside one 1024 x 1024 tile: clipped, no margin noise
            assert np.array_equal(a, np.array(Image.open(os.path.join(d, f))))
            assert tiff_ingest.tiff_info(os.path.join(d, f))["tile_width"] == 1024


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.uint32])
@pytest.mark.parametrize("strips", [False, True])
def test_tiles_and_strips_round_trip(tmp_path, dtype, strips):
    rng = np.random.default_rng(1)
    a = rng.integers(0, np.iinfo(dtype).max, (1500, 2300), dtype=dtype, endpoint=True)      # edges fall inside tiles / strips
    p = str(tmp_path / "a.tif")
    synth.write_tiled_tiff(p, a, tile=1024, strips=strips, rows_per_strip=37)
    b = tiff_ingest.read_tiff(p)
    assert b.dtype == dtype and np.array_equal(a, b)
    i = tiff_ingest.tiff_info(p)
    assert (i["tile_width"] == 0) == strips and i["bits_per_sample"] == a.dtype.itemsize * 8


def _patch_sample_format(path, fmt):
    raw = bytearray(open(path, "rb").read())
    n = struct.unpack_from("<H", raw, 8)[0]
    for k in range(n):
        off = 10 + 12 * k
        if struct.unpack_from("<H", raw, off)[0] == 339:
            struct.pack_into("<H", raw, off + 8, fmt)
    open(path, "wb").write(bytes(raw))


def test_signed_samples_clamp_and_float_is_refused(tmp_path):
    a = np.array([[-5, 7, -1], [300, 0, 32767]], np.int16)
    p = str(tmp_path / "s.tif")
    synth.write_tiled_tiff(p, a.view(np.uint16), tile=16)
    _patch_sample_format(p, 2)                       # SAMPLEFORMAT_INT: same bytes, signed meaning
    b = tiff_ingest.read_tiff(p)
    assert b.tolist() == [[0, 7, 0], [300, 0, 32767]]          # grayscale_tiff.h:283-289: negatives clamp to 0
    f = np.array([[0.5, 1.5]], np.float32)
    q = str(tmp_path / "f.tif")
    synth.write_tiled_tiff(q, f.view(np.uint32), tile=16)
    _patch_sample_format(q, 3)
    with pytest.raises(ValueError, match="floating-point"):
        tiff_ingest.read_tiff(q)
    with pytest.raises(IOError):
        tiff_ingest.read_tiff(str(tmp_path / "missing.tif"))
