"""GPU tests of the size-class launches (nyxhip_api.hip: launch_device_all / run_class).

The reference has no coupling between the ROIs of a batch -- every worker thread takes ROIs of any size
(/root/reference/src/nyx/parallel.h:23-42, roi_cache.h:31-84).  Here a launch is sized by its largest ROI, so a call is split into
launches per size class; these tests pin (1) parity of a heavy-tailed batch against the oracle, (2) that an ROI's row does not
depend on which other ROIs share its call, (3) what the call reports about its classes."""
import numpy as np
import pytest

from nyxus_amd import _abi, _lib
from oracle import pyoracle as po
from tests import parity, synth

pytestmark = pytest.mark.gpu

MASK = _abi.FAM_INTENSITY | _abi.FAM_GLCM
CONFIG4 = MASK | _abi.FAM_GLRLM | _abi.FAM_GLSZM | _abi.FAM_NGTDM


def ellipse_roi(a, b, rng, hi=4096, lo=1, holes=0.0):
    yy, xx = np.mgrid[-b:b + 1, -a:a + 1]
    m = (xx * xx) * (b * b) + (yy * yy) * (a * a) <= (a * a) * (b * b)
    if holes:
        m &= rng.random(m.shape) >= holes
        m[b, a] = True
    y, x = np.nonzero(m)
    o = np.lexsort((y, x))
    return dict(x=x[o], y=y[o], inten=rng.integers(lo, hi, len(x)).astype(np.uint32))


def mixed_rois(seed=1, n=90, big=2):
    """Log-normal radii 2..150 (median 10) + `big` ellipses with 300..400-px boxes + a few wide-range and degenerate ROIs."""
    rng = np.random.default_rng(seed)
    rois = []
    for k in range(n):
        r = int(np.clip(np.rint(np.exp(rng.normal(np.log(10.0), 0.8))), 2, 150))
        hi = 70000 if k % 17 == 5 else 300 if k % 7 == 3 else 4096       # some ranges beyond the 16-bit tables
        rois.append(ellipse_roi(r, max(2, int(r * rng.uniform(0.6, 1.0))), rng, hi=hi, lo=0 if k % 9 == 0 else 1, holes=0.1 if k % 4 == 0 else 0.0))
    for _ in range(big):
        rois.insert(int(rng.integers(0, len(rois))), ellipse_roi(int(rng.integers(150, 200)), int(rng.integers(150, 200)), rng, holes=0.02))
    rois.insert(7, dict(x=[0], y=[0], inten=[9]))                      # single pixel
    rois.insert(11, dict(x=np.arange(40), y=np.zeros(40, int), inten=np.full(40, 5, np.uint32)))   # constant row
    return rois


@pytest.mark.parametrize("mask,gd", [(MASK, 8), (CONFIG4, 8), (MASK, 64), (_abi.FAM_ALL & ~_abi.FAM_GABOR & ~_abi.FAM_SMOMS & ~_abi.FAM_IMOMS, 16)])
def test_mixed_size_batch_matches_oracle(hip_ctx, mask, gd):
    b = _abi.batch_from_rois(mixed_rois())
    s = _abi.default_settings(gd)
    G = hip_ctx.featurize_host(b, mask, s)
    O = po.oracle_featurize(b, mask, s)
    bad = parity.compare_tables(G, O, _lib.column_names(mask, s), batch=b)
    assert not bad, "\n".join(bad[:20])
    rep = hip_ctx.launch_report()
    assert sum(r["rois"] for r in rep) == b.n_roi and len(rep) >= 5, rep          # several classes, every ROI in exactly one
    assert any(r["wide_range"] == 1 for r in rep) and any(r["size_class"] == 4 for r in rep), rep


def test_a_row_does_not_depend_on_its_companions(hip_ctx):
    """The same ROIs alone, and in a call that also holds large, huge and wide-range ROIs: their rows are equal bit for bit
    (class membership, table sizes and summation orders are functions of the ROI and the settings alone)."""
    rng = np.random.default_rng(4)
    small = [ellipse_roi(int(r), int(max(2, r - k % 3)), rng, holes=0.05 * (k % 3)) for k, r in enumerate(rng.integers(2, 31, 40))]
    others = [ellipse_roi(100, 80, rng), ellipse_roi(180, 160, rng), ellipse_roi(12, 12, rng, hi=200000), ellipse_roi(60, 50, rng)]
    s = _abi.default_settings(8)
    mask = _abi.FAM_ALL & ~_abi.FAM_GABOR
    alone = hip_ctx.featurize_host(_abi.batch_from_rois(small), mask, s)
    mixed = hip_ctx.featurize_host(_abi.batch_from_rois(others[:2] + small + others[2:]), mask, s)[2:2 + len(small)]
    names = _lib.column_names(mask, s)
    diff = [(names[j], i) for i, j in zip(*np.nonzero(~((alone == mixed) | (np.isnan(alone) & np.isnan(mixed)))))]
    assert not diff, diff[:10]


def test_launch_report_of_a_hinted_batch_of_small_rois(hip_ctx):
    """Stated extrema within the two smallest size classes: whole-batch launches, nothing counted, no host round trip."""
    b = synth.tile_batch(1)
    s = _abi.default_settings(8)
    hip_ctx.featurize_host(b, MASK, s)
    rep = hip_ctx.launch_report()
    # (round 6: two feature launches on stated extrema beyond the smallest class -- the wave-per-ROI kernel filtered to class 0, then everybody else)
    assert [r["class"] for r in rep] == [-2, -3] and all(r["rois"] == 196 and r["workspace"] == 0 for r in rep), rep
    rois = synth.random_rois(80, seed=3, rmax=20, value_modes=(4096, 256, 8))         # both shape builds (one-wave for the smallest class)
    big = [r for r in rois if len(r["x"]) > 256]
    rois = big + [r for r in rois if len(r["x"]) <= 256][:len(big) // 6]              # (the smallest class rare: see the census test below)
    hip_ctx.featurize_host(_abi.batch_from_rois(rois), MASK | _abi.FAM_ZERNIKE | _abi.FAM_GLSZM, s)
    assert sorted(r["class"] for r in hip_ctx.launch_report()) == [-5, -4, -3, -2, -1]
    rois = synth.random_rois(40, seed=3, rmax=20)                     # value modes up to 2^32 - 1: wide ranges are possible -> exact classes
    hip_ctx.featurize_host(_abi.batch_from_rois(rois), MASK, s)
    assert all(r["class"] >= 0 for r in hip_ctx.launch_report()) and any(r["wide_range"] == 1 for r in hip_ctx.launch_report())


def test_census_of_the_smallest_class_picks_lists_or_filtered_launches(hip_ctx):
    """A batch on stated extrema that mixes the two smallest size classes: filtered whole-batch launches while the smallest class is rare
    (nothing counted, no host round trip), exact class lists once the census of the previous calls says it is common.  Same rows either way;
    a host batch is counted on the host, a device batch by the scanning kernel (read with the status flag at the next sync)."""
    import torch
    rng = np.random.default_rng(12)
    small = [ellipse_roi(int(rng.integers(3, 9)), int(rng.integers(3, 9)), rng) for _ in range(60)]
    mid = [ellipse_roi(int(rng.integers(12, 30)), int(rng.integers(12, 30)), rng) for _ in range(12)]
    s = _abi.default_settings(8)
    dev = torch.device("cuda", 0)

    def device_call(rois):
        b = _abi.batch_from_rois(rois)
        keep = {k: torch.from_numpy(getattr(b, k).view({2: np.int16, 4: np.int32, 8: np.int64}[getattr(b, k).dtype.itemsize])).to(dev)
                for k in ("px_offset", "x", "y", "inten", "bbox_w", "bbox_h", "min_inten", "max_inten")}
        cb = b.c_struct()
        for k, t in keep.items():
            setattr(cb, k, t.data_ptr())
        cb.slide_min = None; cb.slide_max = None; cb.memory = _abi.MEM_DEVICE
        ncol = hip_ctx.n_columns(MASK, s)
        out = torch.full((b.n_roi, ncol), -1.0, dtype=torch.float64, device=dev)
        hip_ctx.featurize_device_async(cb, MASK, s, out.data_ptr(), ncol)
        rep = [r["class"] for r in hip_ctx.launch_report()]
        hip_ctx.sync()
        return out.cpu().numpy(), rep, b
    # a device batch of class 1 only resets the census (its scan meets no small ROI)
    device_call(mid)
    T1, rep1, b = device_call(small + mid)          # census: nothing small seen -> filtered launches; their scan counts 60 of 72
    T2, rep2, _ = device_call(small + mid)          # -> exact lists
    assert rep1 == [-2, -3] and all(c >= 0 for c in rep2) and 0 in rep2, (rep1, rep2)
    assert np.array_equal(T1.view(np.uint64), T2.view(np.uint64))
    assert not parity.compare_tables(T1, po.oracle_featurize(b, MASK, s), _lib.column_names(MASK, s), batch=b)
    T3, rep3, _ = device_call(mid + small[:2])      # the lists' headers said 60 of 72: still lists; they now say 2 of 14
    T4, rep4, _ = device_call(mid + small[:2])      # -> filtered launches again
    assert all(c >= 0 for c in rep3) and rep4 == [-2, -3], (rep3, rep4)
    assert np.array_equal(T3.view(np.uint64), T4.view(np.uint64))
    # host batches are counted on the host: the right form at once
    hip_ctx.featurize_host(_abi.batch_from_rois(small + mid), MASK, s)
    assert all(r["class"] >= 0 for r in hip_ctx.launch_report())
    hip_ctx.featurize_host(_abi.batch_from_rois(mid + small[:2]), MASK, s)
    assert [r["class"] for r in hip_ctx.launch_report()] == [-2, -3]


def test_a_wrong_statement_about_the_batch_is_an_error(hip_ctx):
    """max_px / max_bbox_side stated smaller than an ROI of the batch: the call fails (error flag raised by the classifier) instead of
    missing the ROI or overrunning a carve-out."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(2)
    b = _abi.batch_from_rois([ellipse_roi(5, 5, rng), ellipse_roi(40, 30, rng), ellipse_roi(7, 6, rng)])
    s = _abi.default_settings(8)
    dev = torch.device("cuda", 0)
    keep = {k: torch.from_numpy(getattr(b, k).view({2: np.int16, 4: np.int32, 8: np.int64}[getattr(b, k).dtype.itemsize])).to(dev)
            for k in ("px_offset", "x", "y", "inten", "bbox_w", "bbox_h", "min_inten", "max_inten")}
    cb = b.c_struct()
    for k, t in keep.items():
        setattr(cb, k, t.data_ptr())
    cb.slide_min = None; cb.slide_max = None; cb.memory = _abi.MEM_DEVICE
    ncol = hip_ctx.n_columns(MASK, s)
    out = torch.empty((b.n_roi, ncol), dtype=torch.float64, device=dev)
    hip_ctx.featurize_device_async(cb, MASK, s, out.data_ptr(), ncol)          # the true extrema: fine
    hip_ctx.sync()
    cb.max_px = 200; cb.max_bbox_area = 400; cb.max_bbox_side = 20                # a wrong statement
    hip_ctx.featurize_device_async(cb, MASK, s, out.data_ptr(), ncol)
    with pytest.raises(_lib.NyxHipError) as ei:
        hip_ctx.sync()
    assert ei.value.code == 5
    # the same wrong statement with a shape-only mask (the hinted path launches the shape kernels over the whole batch, filtered by
    # class: an ROI outside the stated classes must not be skipped by both launches -- round-3 advisor).  Either the flag is raised
    # (Gabor: the ROI's plane does not fit the carve-out the statement sized) or the rows are right all the same (Zernike keeps no
    # ROI-sized state: an ROI above its staging cap is read from HBM) -- never silence with unwritten columns.
    for shape_mask in (_abi.FAM_ZERNIKE, _abi.FAM_GABOR, _abi.FAM_ZERNIKE | _abi.FAM_GABOR):
        nc2 = hip_ctx.n_columns(shape_mask, s)
        out2 = torch.full((b.n_roi, nc2), -12345.0, dtype=torch.float64, device=dev)
        hip_ctx.featurize_device_async(cb, shape_mask, s, out2.data_ptr(), nc2)
        try:
            hip_ctx.sync()
        except _lib.NyxHipError as e2:
            assert e2.code == 5, shape_mask
            assert shape_mask & _abi.FAM_GABOR
        else:
            assert shape_mask == _abi.FAM_ZERNIKE
            assert not parity.compare_tables(out2.cpu().numpy(), po.oracle_featurize(b, shape_mask, s), _lib.column_names(shape_mask, s))
    cb.max_px = cb.max_bbox_area = cb.max_bbox_side = 0                           # no statement: the classifier derives everything
    hip_ctx.featurize_device_async(cb, MASK, s, out.data_ptr(), ncol)
    hip_ctx.sync()
    want = po.oracle_featurize(b, MASK, s)
    assert not parity.compare_tables(out.cpu().numpy(), want, _lib.column_names(MASK, s))


def _n_gpus():
    import torch
    return torch.cuda.device_count()          # (counting devices does not initialise them)


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs")
def test_two_gpus_bench_and_sharded_api_agree_with_one():
    """The moment two devices exist: `bench.py --gpus 2` over RCCL (per-rank devices distinct, table gathered, parity gate ok) and
    Nyxus(gpu_devices=[0, 1]) equal to the single-device result bit for bit.  (The driver's 8-GPU SCALE run is the same code.)"""
    import json
    import os
    import subprocess
    import sys
    import nyxus_amd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--tiles", "16", "--steps", "2", "--warmup", "1", "--no-extras",
                        "--tile-path-tiles", "0", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert rec["n_gpus"] == 2 and len(rec["per_rank"]) == 2 and "error" not in rec
    assert len({r["uuid"] or r["pci_bus_id"] for r in rec["per_rank"]}) == 2
    assert rec["table_gather"]["error"] is None and "ok" in rec["config"]["parity_check"] and "MISMATCH" not in rec["config"]["parity_check"]
    rng = np.random.default_rng(8)
    I = rng.integers(1, 4096, (6, 128, 128)).astype(np.uint16)
    M = np.stack([synth.disk_label_tile(size=128, pitch=32, radius=6 + 2 * k) * (k + 1) for k in range(6)]).astype(np.uint32)
    feats = ["*ALL_INTENSITY*", "*ALL_GLCM*", "*ALL_GLSZM*"]
    one = nyxus_amd.Nyxus(feats, coarse_gray_depth=8).featurize(I, M)
    two = nyxus_amd.Nyxus(feats, coarse_gray_depth=8, gpu_devices=[0, 1]).featurize(I, M)
    assert one.equals(two)


# ---- the smallest size class on its own kernel: a wave per ROI (roi_small.hip, round 6) -------------------------------------------------
def _small_rois(seed, hi=4096):
    """ROIs of the smallest size class (<= 256 px, sides <= 32) around every boundary the wave-per-ROI kernel has: 1, 2, 63, 64, 65, 127, 128,
    129, 255, 256 pixels (one / four keys per lane, the three sort widths), zeros, constants, a blank ROI, two values only, ties at the
    percentile boundaries, thin strips, and shapes with holes."""
    rng = np.random.default_rng(seed)
    rois = []
    for n in (1, 2, 3, 5, 31, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256):
        w = int(min(32, max(1, np.ceil(np.sqrt(n)))))
        h = int(np.ceil(n / w))
        if h > 32:
            w, h = 32, int(np.ceil(n / 32))
        yy, xx = np.divmod(np.arange(n), w)
        rois.append(dict(x=xx, y=yy, inten=rng.integers(0 if n % 3 == 0 else 1, hi, n).astype(np.uint32)))
    for r in (3, 4, 6, 8, 9):
        rois.append(ellipse_roi(r, r, rng, hi=hi))
        rois.append(ellipse_roi(r, max(2, r - 2), rng, hi=hi, lo=0, holes=0.15))
    c = ellipse_roi(6, 5, rng); c["inten"][:] = 1234; rois.append(c)                       # constant
    z = ellipse_roi(5, 5, rng); z["inten"][:] = 0; rois.append(z)                          # blank
    t = ellipse_roi(8, 7, rng); t["inten"][:] = np.where(np.arange(len(t["inten"])) % 2, 10, 3000); rois.append(t)   # two values
    q = ellipse_roi(7, 7, rng); q["inten"][:] = (np.arange(len(q["inten"])) // 4 * 25 + 100).astype(np.uint32); rois.append(q)   # ties
    rois.append(dict(x=np.arange(32), y=np.zeros(32, int), inten=rng.integers(1, hi, 32).astype(np.uint32)))           # a row
    rois.append(dict(x=np.zeros(30, int), y=np.arange(30), inten=rng.integers(1, 300, 30).astype(np.uint32)))          # a column
    rois.append(dict(x=[0], y=[0], inten=[7]))
    return rois


@pytest.mark.parametrize("mask", [1, 2, 3])
@pytest.mark.parametrize("gd,hi", [(8, 4096), (8, 256), (16, 16000), (3, 50), (64, 4096), (64, 40), (33, 1000), (17, 300)])
def test_smallest_class_matches_oracle(hip_ctx, mask, gd, hi):
    """Every ROI of the batch is of class 0: a whole-batch launch of the wave-per-ROI kernel on stated extrema (17..64 levels: the GLCM
    features come from the pairs of the ROI, not from the 64 x 64 matrix)."""
    s = _abi.default_settings(gd)
    b = _abi.batch_from_rois(_small_rois(gd + hi, hi=hi))
    G = hip_ctx.featurize_host(b, mask, s)
    O = po.oracle_featurize(b, mask, s)
    bad = parity.compare_tables(G, O, _lib.column_names(mask, s), batch=b)
    assert not bad, "\\n".join(bad[:20])


@pytest.mark.parametrize("gd", [8, 64])
def test_smallest_class_rows_do_not_depend_on_companions_or_options(hip_ctx, gd):
    """A class-0 row is the same alone, among larger companions (filtered whole-batch launches / exact lists), with other families in
    the call, with slide extrema given -- and symmetric counts, angle subsets, other offsets match the oracle."""
    rng = np.random.default_rng(77)
    small = _small_rois(5)
    s = _abi.default_settings(gd)
    alone = hip_ctx.featurize_host(_abi.batch_from_rois(small), 3, s)
    mid = [ellipse_roi(int(rng.integers(12, 30)), int(rng.integers(12, 30)), rng) for _ in range(12)]          # size class 1
    big = [ellipse_roi(90, 70, rng)]                                                                            # a class that forces the exact path
    for others in (mid, mid + big):
        mixed = []
        for i, r in enumerate(small):
            mixed.append(r)
            if i < len(others): mixed.append(others[i])
        idx = [i for i, r in enumerate(mixed) if any(r is q for q in small)]
        bm = _abi.batch_from_rois(mixed)
        for m in (3, 3 | _abi.FAM_GLRLM | _abi.FAM_ZERNIKE):
            T = hip_ctx.featurize_host(bm, m, s)
            names = _lib.column_names(m, s)
            keep = [i for i, nme in enumerate(names) if nme in _lib.column_names(3, s)]
            assert np.array_equal(T[idx][:, keep].view(np.uint64), alone.view(np.uint64))
            bad = parity.compare_tables(T, po.oracle_featurize(bm, m, s), names, batch=bm)
            assert not bad, "\\n".join(bad[:10])
    for sym, angles, offset in ((1, (0, 45, 90, 135), 1), (0, (45, 135), 1), (1, (135, 0, 90), 1), (1, (90,), 2), (0, (0, 45, 90, 135), 3)):
        s2 = _abi.default_settings(gd)
        s2.glcm_symmetric = sym; s2.glcm_offset = offset; s2.glcm_n_angles = len(angles)
        for i, a in enumerate(angles): s2.glcm_angles[i] = a
        b = _abi.batch_from_rois(small)
        bad = parity.compare_tables(hip_ctx.featurize_host(b, 3, s2), po.oracle_featurize(b, 3, s2), _lib.column_names(3, s2), batch=b)
        assert not bad, "\\n".join(bad[:10])
