#!/usr/bin/env python3
"""bench.py -- ROIs/sec of the per-ROI feature reduce (*ALL_GLCM* + *ALL_INTENSITY*)
on synthetic 1024x1024 tiles, the metric of BASELINE.json.

A "step" is one pass of the hot path (nyxhip_featurize_batch_async: the replacement
of reduce_trivial_rois_manual, /root/reference/src/nyx/reduce_trivial_rois.cpp:772-795)
over one batch of ROIs that is already resident in HBM: `--tiles` tiles (default 1000,
the tile count of BASELINE.json configs[1]/[2]) x 196 ROIs per tile (14x14 disks of
radius 30 -> 2821 px, bbox 61x61), intensities uniform in [1, 4095],
coarse_gray_depth=8, GLCM angles {0,45,90,135}, offset 1, matlab binning.

Per rank (one process per GPU): the same number of tiles (weak scaling), no collective in
the timed region; the one exchange of a job -- the gather of the final feature table to
rank 0 (RCCL) -- runs once after the timed steps and is reported as `table_gather`.
Output: ONE JSON line on rank 0 (contract in the task statement) plus `roofline` and
`cpu_baseline` objects.

Launching: `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment makes
this process a launcher: it never touches the GPU, starts N child ranks of this same script
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set) and exits with the
worst child status.  Under torchrun (WORLD_SIZE already set) every process is a rank.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
FP64_PEAK_TF = 78.6    # AMD Instinct MI355X data sheet: peak fp64 vector = fp64 matrix, 78.6 TFLOP/s (the guide has no fp64 row)
FP32_PEAK_TF = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: peak FP32 (vector) 157.3 TFLOPS
F16_MFMA_PEAK_TF = 2500.0   # same guide: BF16 / F16 MFMA ~2.5 PFLOP/s dense


def gabor_roofline(alg_flops, dt, bbox_w, bbox_h, max_inten, n_filters):
    """The Gabor kernel against the pipe that executes it: `achieved` = algorithmic flops / time against the dense f16 MFMA peak; beside
    it the flops the MFMA stage issues: (w / 4 + 1) column groups x ceil(h / 16) row tiles x 8 tap-row pairs x 4 columns x digit planes
    (1 below 2^11, 2 below 2^16) x groups of four filters, 2 * 16 * 16 * 32 flops each."""
    w = np.asarray(bbox_w, np.float64); h = np.asarray(bbox_h, np.float64); mx = np.asarray(max_inten, np.float64)
    units = (np.floor(w / 4) + 1) * np.ceil(h / 16)
    digits = np.where(mx < 2048, 1.0, 2.0)
    on_mfma = mx < 65536
    ex = float(np.sum(units * 32.0 * digits * np.ceil(n_filters / 4.0) * 16384.0 * on_mfma))
    return {"bound": "mfma", "achieved": alg_flops / dt / 1e12, "peak": F16_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": alg_flops / dt / 1e12 / F16_MFMA_PEAK_TF,
            "algorithmic_flops_per_launch": alg_flops, "executed_mfma_flops_per_launch": ex, "executed_mfma_TFLOPs": ex / dt / 1e12}



def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--tiles", type=int, default=1000, help="tiles per GPU per step")
    ap.add_argument("--gray-depth", type=int, default=8)
    ap.add_argument("--cpu-tiles", type=int, default=0, help="tiles in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--families", type=int, default=3, help="family bitmask (diagnostic; the metric is 3 = INTENSITY|GLCM)")
    ap.add_argument("--tile-path-tiles", type=int, default=1000, help="tiles for the informational fused tile-path measurement (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip the informational legs (grey depth 64, configs 4 and 5)")
    ap.add_argument("--stub", action="store_true",
                    help="TEST HOOK (tests/test_bench_launcher.py): no GPU, gloo; every rank fills a rank-coded table on the CPU so that "
                         "the launcher, the barrier / max-over-ranks timing and the table gather can run here; the line says \"stub\": true")
    return ap.parse_args()


def under_profiler() -> bool:
    """rocprofv3 preloads its tool library into this process, which initialises the GPU before main() runs (with --pmc it
    certainly does): starting child ranks from here would be an exec out of a GPU-initialised process -- forbidden on this pool."""
    pre = os.environ.get("LD_PRELOAD", "")
    return any(t in pre for t in ("rocprof", "roctracer", "rocprofiler")) or any(k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER_")) for k in os.environ)


def launch(a) -> int:
    """`--gpus N` without a rendezvous in the environment: start N ranks of this script, one per GPU.
    The launcher itself never initialises HIP (children are fresh processes, nothing is exec'ed over a GPU process) -- which
    does not hold under a profiler preload: profiling is single-rank only."""
    if under_profiler():
        print("bench.py: --gpus > 1 cannot be launched from a profiled process (the profiler has initialised the GPU here); "
              "profile one rank (--gpus 1)", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            c = p.poll()
            if c is None:
                continue
            pending.remove(p)
            if c != 0:
                rc = rc or c
                for q in pending:                 # one rank failed: the others would wait in a collective forever
                    q.terminate()
        time.sleep(0.05)
    return rc


def run_stub(a, world, rank):
    """Test hook: the distributed skeleton of the bench (shard ranges, fence, max-over-ranks time, table gather) on gloo."""
    import torch
    import torch.distributed as dist
    from nyxus_amd.sharding import TableGather, shard_range
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ncol, rois_per_tile = 4, 3
    if a.tiles < 0:
        raise ValueError("--tiles must be >= 0")
    lo, hi = shard_range(a.tiles * world, rank, world)          # weak scaling: a.tiles per rank
    rows = torch.tensor([[t * 1000.0 + r + 0.001 * c for c in range(ncol)] for t in range(lo, hi) for r in range(rois_per_tile)],
                        dtype=torch.float64).reshape(-1, ncol)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        last = rows.clone()
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    mine = {"rank": rank, "device": "cpu-stub", "ms_per_step": 1e3 * float(el.item()) / max(a.steps, 1), "rows": int(rows.shape[0])}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    g = TableGather(ncol, dst=0)
    g.start(last)
    full = g.finish()
    gate_rc = 0
    if rank == 0:
        want = [t * 1000.0 + r for t in range(a.tiles * world) for r in range(rois_per_tile)]
        if os.environ.get("NYX_BENCH_BREAK_GATE") and full.shape[0]:     # TEST HOOK: a wrong table must cost the value and the exit code
            full[0, 0] += 1.0
        ok = full.shape[0] == len(want) and bool((full[:, 0].floor() == torch.tensor(want, dtype=torch.float64)).all())
        rec = {"metric": "stub", "stub": True, "value": float(full.shape[0]), "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": 1e3 * float(el.item()) / max(a.steps, 1), "rows": int(full.shape[0]), "row_order_ok": ok, "per_rank": per_rank,
               "config": {"workload": "stub", "parity_check": "ok" if ok else "1 MISMATCHES (stub rows)"}}
        gate_rc = apply_gates(rec)                   # the same gate the real line goes through
        print(json.dumps(rec))
    if world > 1:
        grc = torch.tensor([gate_rc], dtype=torch.int32)
        dist.broadcast(grc, src=0)
        gate_rc = int(grc.item())
        dist.barrier()
        dist.destroy_process_group()
    if gate_rc:
        sys.exit(gate_rc)


def disk_cloud(radius=30):
    """Column-major cloud of one benchmark ROI (phase2_2d.cpp:655-656 scan order)."""
    r = radius
    yy, xx = np.mgrid[-r:r + 1, -r:r + 1]
    m = (xx * xx + yy * yy) <= r * r
    y, x = np.nonzero(m)
    o = np.lexsort((y, x))
    return x[o].astype(np.uint16), y[o].astype(np.uint16), 2 * r + 1


def host_rows(dev_arrays, idx):
    """HostBatch of the ROIs `idx` of a device-resident batch (dict of torch tensors: px_offset, x, y, inten, bbox_w, bbox_h,
    min_inten, max_inten) -- what the oracle is given in the parity gates."""
    from nyxus_amd import _abi
    off = dev_arrays["px_offset"].cpu().numpy().astype(np.int64)
    idx = np.asarray(idx, np.int64)
    segs = [np.arange(off[i], off[i + 1]) for i in idx]
    cat = np.concatenate(segs)
    sel = lambda k, dt: dev_arrays[k].cpu().numpy().view(dt)
    import torch
    ct = torch.from_numpy(cat).to(dev_arrays["x"].device)
    g = lambda k, dt: dev_arrays[k][ct].cpu().numpy().view(dt)
    no = np.concatenate([[0], np.cumsum([len(s_) for s_ in segs])]).astype(np.uint64)
    return _abi.HostBatch(np.asarray(idx + 1, np.uint32), no, g("x", np.uint16), g("y", np.uint16), g("inten", np.uint32), sel("bbox_w", np.uint32)[idx],
                          sel("bbox_h", np.uint32)[idx], sel("min_inten", np.uint32)[idx], sel("max_inten", np.uint32)[idx])


def margins(got, want, names):
    """SURVEY.md 8(d) metric row: max relative error vs the CPU path per feature column.  Returns the five largest
    {column: max over rows of |got - want| / |want|} (rows where both are finite and want != 0) and how many columns are equal bit for bit."""
    got = np.asarray(got, np.float64); want = np.asarray(want, np.float64)
    same = (got == want) | (np.isnan(got) & np.isnan(want))
    n_exact = int(same.all(axis=0).sum())
    with np.errstate(invalid="ignore", divide="ignore"):
        rel = np.where(same | ~np.isfinite(got) | ~np.isfinite(want) | (want == 0), 0.0, np.abs(got - want) / np.abs(want))
    col = rel.max(axis=0) if rel.shape[0] else np.zeros(rel.shape[1])
    top = np.argsort(-col)[:5]
    return {"top5": {names[j]: float(col[j]) for j in top if col[j] > 0}, "exact_columns": n_exact, "columns": int(got.shape[1]), "rows": int(got.shape[0])}


def gate(table_rows, hb, mask, s, marg=None):
    """Parity gate of a timed leg (a timed configuration counts only if its features match): rows of the
    table the leg produced vs the CPU oracle on the same ROIs.  Returns "ok" or the mismatch count (details on stderr);
    marg (a dict) receives the leg's per-column error margins."""
    from nyxus_amd import _lib
    from oracle import pyoracle as po
    from tests import parity
    want = po.oracle_featurize(hb, mask, s)
    names = _lib.column_names(mask, s)
    got = np.asarray(table_rows)
    if os.environ.get("NYX_BENCH_BREAK_GATE"):       # TEST HOOK: a deliberately wrong table must fail the bench
        got = got.copy(); got[0, 0] += 1.0
    bad = parity.compare_tables(got, want, names, batch=hb)
    if marg is not None:
        marg.update(margins(got, want, names))
    if bad:
        print("\n".join(bad[:10]), file=sys.stderr)
    return "ok" if not bad else f"{len(bad)} MISMATCHES"


def gate_ok(v) -> bool:
    """A leg's parity_check string (or None: not checked) says the leg counts."""
    return v is None or (isinstance(v, str) and "ok" in v and "MISMATCH" not in v and "FAILED" not in v)


def apply_gates(rec) -> int:
    """a timed configuration counts only if its features match.  Every object of the line that carries a
    `parity_check` which is not ok loses its `value` (null) and the process exits with code 4 after printing the line.
    Returns the exit code (0 or 4)."""
    failed = []

    def walk(obj, path):
        if isinstance(obj, dict):
            pc = obj.get("parity_check", "absent")
            if pc != "absent" and not gate_ok(pc):
                failed.append(path or "headline")
                for k in ("value", "rois_per_s", "ns_per_roi", "ms_per_call", "ms_per_step"):
                    if k in obj:
                        obj[k] = None
            for k, v in obj.items():
                walk(v, f"{path}.{k}" if path else k)
        elif isinstance(obj, list):
            for i, v in enumerate(obj):
                walk(v, f"{path}[{i}]")
    cfg = rec.get("config", {})
    if "parity_check" in cfg and not gate_ok(cfg["parity_check"]):
        failed.append("headline")
        rec["value"] = None
    walk({k: v for k, v in rec.items() if k != "config"}, "")
    if failed:
        rec["error"] = (rec.get("error", "") + "; " if rec.get("error") else "") + "parity gate failed: " + ", ".join(sorted(set(failed)))
        return 4
    return 0


class Bench:
    """One rank of the bench: the resident batch, the timed steps, and one method per leg of the line (every leg fills its own
    key of `rec`; an informational leg that fails records {"error": ...} and never costs the headline)."""
    ROIS_PER_TILE = 196

    def __init__(self, a, world, rank, local_rank):
        import torch
        import torch.distributed as dist
        from nyxus_amd import _abi, _lib
        from nyxus_amd.sharding import TableGather
        self.a, self.world, self.rank, self.local_rank = a, world, rank, local_rank
        self.torch, self.dist, self._abi, self._lib = torch, dist, _abi, _lib
        if world > 1:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)
        if world > 1:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=self.dev)
        self.mask = a.families
        self.s = _abi.default_settings(a.gray_depth)
        self.ctx = _lib.Context(local_rank)
        self.ncol = self.ctx.n_columns(self.mask, self.s)
        self.gather = TableGather(self.ncol, dst=0) if world > 1 else None
        self.rec = None
        self.exact_cache = {}

    # ---- synthetic batch, generated on the device (seeded) ---------------------------------
    def make_batch(self):
        torch, a, dev, _abi = self.torch, self.a, self.dev, self._abi
        px, py, side = disk_cloud(30)
        self.side = side
        self.n_px_roi = len(px)                  # 2821
        self.n_roi = a.tiles * self.ROIS_PER_TILE
        self.n_px = self.n_roi * self.n_px_roi
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(1234 + self.rank)
        n_roi, n_px_roi = self.n_roi, self.n_px_roi
        self.inten = torch.randint(1, 4096, (self.n_px,), generator=self.gen, device=dev, dtype=torch.int32)
        self.x = torch.from_numpy(px.view(np.int16)).to(dev).repeat(n_roi)
        self.y = torch.from_numpy(py.view(np.int16)).to(dev).repeat(n_roi)
        self.off = torch.arange(0, n_roi + 1, device=dev, dtype=torch.int64) * n_px_roi
        self.bw = torch.full((n_roi,), side, device=dev, dtype=torch.int32)
        self.bh = torch.full((n_roi,), side, device=dev, dtype=torch.int32)
        iv = self.inten.view(n_roi, n_px_roi)
        self.mn = iv.min(dim=1).values.contiguous()
        self.mx = iv.max(dim=1).values.contiguous()
        self.labels = (torch.arange(n_roi, device=dev, dtype=torch.int32) % self.ROIS_PER_TILE) + 1
        self.outs = [torch.empty((n_roi, self.ncol), dtype=torch.float64, device=dev) for _ in range(2)]
        cb = _abi.Batch()
        cb.n_roi = n_roi
        cb.roi_label = self.labels.data_ptr(); cb.px_offset = self.off.data_ptr()
        cb.x = self.x.data_ptr(); cb.y = self.y.data_ptr(); cb.inten = self.inten.data_ptr()
        cb.bbox_w = self.bw.data_ptr(); cb.bbox_h = self.bh.data_ptr()
        cb.min_inten = self.mn.data_ptr(); cb.max_inten = self.mx.data_ptr()
        cb.slide_min = None; cb.slide_max = None
        cb.memory = _abi.MEM_DEVICE
        cb.max_px = n_px_roi; cb.max_bbox_area = side * side
        cb.max_inten_range = int((self.mx - self.mn).max().item())
        cb.max_bbox_side = side
        self.cb = cb
        self.dev_arrays = {"px_offset": self.off, "x": self.x, "y": self.y, "inten": self.inten, "bbox_w": self.bw, "bbox_h": self.bh,
                           "min_inten": self.mn, "max_inten": self.mx}
        self.last_tile = np.arange((a.tiles - 1) * self.ROIS_PER_TILE, a.tiles * self.ROIS_PER_TILE)
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)       # kernels + events on torch's current stream

    def fence(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    # The path shards by ROI with no exchange step: the timed region holds no collective.  The one collective of a
    # multi-GPU job -- the gather of the final feature table to rank 0 (north_star) -- runs once after the timed steps and
    # is reported on its own (`table_gather`): at 290 MB of table per 3.5 ms step a per-step gather would measure the
    # xGMI link (~60 GB/s per peer), not the reduce path.
    def timed_steps(self):
        a, ctx = self.a, self.ctx

        def step(i):
            out = self.outs[i & 1]
            ctx.featurize_device_async(self.cb, self.mask, self.s, out.data_ptr(), self.ncol)
            return out
        for i in range(a.warmup):
            step(i)
        self.fence()
        # roofline.achieved: the launch group's device time from TWO HIP events on the launch stream (the library launches on torch's
        # current stream, make_batch) around the K timed steps.  (The library's own hooks -- nyxhip_timing_enable -- record two events per
        # call: each is a marker packet the command processor handles between kernels, ~15 us per step boundary = 0.7 % of the headline.)
        e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        last = None
        for i in range(a.steps):
            last = step(i)
        e1.record()
        self.fence()
        t1 = time.perf_counter()
        ctx.sync()                               # raises on a device-side error flag
        self.kern_ms, self.n_launch = e0.elapsed_time(e1) / max(a.steps, 1), a.steps
        self.last, self.t_steps = last, t1 - t0

    def ranks_and_gather(self):
        """Per-rank detail for the one line the driver keeps (which physical device each rank ran on -- two ranks on one device would
        still print a plausible aggregate), the max-over-ranks time, and the job's one collective: the table gather."""
        torch, dist, a, world, rank = self.torch, self.dist, self.a, self.world, self.rank
        props = torch.cuda.get_device_properties(self.local_rank)
        pci = getattr(props, "pci_bus_id", None)
        try:
            dev_uuid = str(props.uuid)
        except Exception:
            dev_uuid = None
        mine = {"rank": rank, "local_rank": self.local_rank, "device": torch.cuda.current_device(), "name": props.name, "pci_bus_id": pci, "uuid": dev_uuid,
                "ms_per_step": 1e3 * self.t_steps / max(a.steps, 1), "kernel_ms": self.kern_ms, "rows": self.n_roi}
        self.per_rank = [mine]
        if world > 1:
            self.per_rank = [None] * world
            dist.all_gather_object(self.per_rank, mine)
        self.same_device = None
        ids = [(r.get("uuid") or r.get("pci_bus_id") or r.get("device")) for r in self.per_rank]
        if len(set(ids)) != len(ids):
            self.same_device = f"ranks share a device: {ids}"
        elapsed = torch.tensor([self.t_steps], dtype=torch.float64, device=self.dev)
        if world > 1:
            dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        self.elapsed = float(elapsed.item())
        self.gather_ms = self.gather_err = None
        if self.gather is not None:              # final table -> rank 0, once, outside the timed region
            try:
                g0 = time.perf_counter()
                self.gather.start(self.last, rows_per_rank=[self.n_roi] * world, producer=self.ctx)
                full = self.gather.finish()
                self.fence()
                self.gather_ms = 1e3 * (time.perf_counter() - g0)
                if rank == 0 and full is not None and tuple(full.shape) != (self.n_roi * world, self.ncol):
                    self.gather_err = f"gathered table has shape {tuple(full.shape)}"
                del full
            except Exception as e:               # the bench line is still printed; the failure is part of it
                self.gather_err = repr(e)

    def hbm_traffic(self):
        """PMC-derived HBM bytes per launch of the metric kernels (profiles/README.md): replayed from profiles/hbm_traffic.json, valid
        only for the kernel sources it was measured on (keyed by their hash) -- and labelled as such (`traffic_source`)."""
        a = self.a
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if not os.path.exists(tpath):
            return None, None
        try:
            import hashlib
            tj = json.load(open(tpath))
            src = hashlib.sha256(b"".join(open(os.path.join(ROOT, "nyxus_amd", "csrc", f_), "rb").read()     # every source the metric kernels are built from
                                            for f_ in ("roi_features.hip", "glcm_rows.h", "device_math.h", "roi_kernel.h"))).hexdigest()
            if tj.get("tiles") == a.tiles and tj.get("gray_depth") == a.gray_depth and tj.get("kernel_source_sha256") == src:
                return tj.get("hbm_bytes_per_launch"), "replayed from profiles/hbm_traffic.json@" + str(tj.get("round", tj.get("measured", "?"))) + \
                    " (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE of this very kernel source, separate passes; not measured in this run)"
        except Exception:
            pass
        return None, None

    def headline(self):
        a, world = self.a, self.world
        n_roi, ncol, n_px = self.n_roi, self.ncol, self.n_px
        total_rois = n_roi * world * a.steps
        self.value = total_rois / self.elapsed
        # algorithmic bytes per launch (pre-assembled clouds):
        # 8 B per ROI pixel in (x:u16, y:u16, inten:u32) + 8 B x n_cols out per ROI
        alg_bytes = n_px * 8 + n_roi * ncol * 8
        achieved = alg_bytes / (self.kern_ms * 1e-3) / 1e9 if self.kern_ms > 0 else 0.0
        traffic, traffic_source = self.hbm_traffic()
        rec = {
            "metric": "ROIs/sec (*ALL_GLCM*+*ALL_INTENSITY*, 1024^2 tiles)",
            "value": self.value, "unit": "ROIs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * self.elapsed / a.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"*ALL_GLCM*+*ALL_INTENSITY*, coarse_gray_depth={a.gray_depth}, 4 angles, d=1, "
                                   f"{a.tiles} synthetic 1024x1024 tiles/GPU x {self.ROIS_PER_TILE} ROIs/tile "
                                   f"(disk r=30, {self.n_px_roi} px, bbox {self.side}x{self.side}), intensities U[1,4095]; "
                                   "reduce stage on pre-assembled ROI clouds resident in HBM",
                       "rois_per_step_per_gpu": n_roi, "n_columns": ncol,
                       "sharding": f"{world} rank(s), tiles block-partitioned, no collective in the timed region; "
                                   "final table gathered to rank 0 over RCCL afterwards (table_gather)"
                                   if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "roi_features_kernel (+ glcm_features_kernel: one launch group, timed together)" if self.mask == 3 else "reduce launch group",
                         "kernel_ms": self.kern_ms, "launches": int(self.n_launch),
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        rec["per_rank"] = self.per_rank
        if self.same_device:
            rec["error"] = self.same_device
        if self.gather_ms is not None or self.gather_err is not None:
            gb = n_roi * world * ncol * 8 / 1e9
            rec["table_gather"] = {"ms": self.gather_ms, "GB": gb, "GBps": (gb / (self.gather_ms * 1e-3)) if self.gather_ms else None, "error": self.gather_err,
                                   "what": "one RCCL gather of the last step's table (all ranks -> rank 0), outside the timed region"}
        self.rec = rec

    def all_rows_invariant(self, table, msk, st):
        """EVERY row of a table computed from the resident batch against exact device reductions of the same arrays
        (MIN / MAX / RANGE / MEAN / INTEGRATED_INTENSITY are integer-exact columns): a defect at large ROI indices cannot hide
        behind the three tiles the oracle sees."""
        torch = self.torch
        cols = self._lib.column_names(msk, st)
        if "MIN" not in cols:
            return "n/a"
        if not self.exact_cache:
            iv64 = self.inten.view(self.n_roi, self.n_px_roi).to(torch.int64)
            tot = iv64.sum(dim=1).to(torch.float64)
            # (tensor / tensor: a Python-scalar divisor makes torch multiply by the reciprocal, which rounds differently from the division)
            self.exact_cache.update({"MIN": self.mn.to(torch.float64), "MAX": self.mx.to(torch.float64), "INTEGRATED_INTENSITY": tot,
                                     "MEAN": torch.div(tot, torch.full_like(tot, float(self.n_px_roi))), "RANGE": (self.mx - self.mn).to(torch.float64)})
            del iv64
        wrong = {c: int((table[:, cols.index(c)] != v).sum().item()) for c, v in self.exact_cache.items()}
        return "ok" if not any(wrong.values()) else "MISMATCHES " + json.dumps({c: k for c, k in wrong.items() if k})

    def headline_gate(self):
        """Parity gate of what was timed: first, middle and LAST tile of the last step vs the oracle, and every row of the table
        against exact reductions of the same device arrays (a defect at large ROI indices cannot hide)."""
        a, k = self.a, self.ROIS_PER_TILE
        res = []
        marg_h = {}
        for t in sorted({0, a.tiles // 2, a.tiles - 1}):
            idx = np.arange(t * k, (t + 1) * k)
            res.append(gate(self.last[idx[0]:idx[-1] + 1].cpu().numpy(), host_rows(self.dev_arrays, idx), self.mask, self.s, marg=marg_h))
        self.rec["config"]["max_rel_err"] = marg_h          # (of the last tile checked)
        inv = self.all_rows_invariant(self.last, self.mask, self.s)
        ok = all(r == "ok" for r in res) and inv in ("ok", "n/a")
        self.rec["config"]["parity_check"] = (("" if ok else "FAILED: ") + f"tiles 0, {a.tiles // 2}, {a.tiles - 1} of the last timed step vs oracle: " + "/".join(res)
                                              + f"; all {self.n_roi} rows, MIN/MAX/RANGE/MEAN/INTEGRATED_INTENSITY vs exact device reductions: {inv}")

    def host_slice(self, k):
        """HostBatch of the first k ROIs of the resident batch (one contiguous slice of every array)."""
        n = self.n_px_roi
        return self._abi.HostBatch(self.labels[:k].cpu().numpy().astype(np.uint32), self.off[:k + 1].cpu().numpy().astype(np.uint64),
                                   self.x[:k * n].cpu().numpy().view(np.uint16), self.y[:k * n].cpu().numpy().view(np.uint16),
                                   self.inten[:k * n].cpu().numpy().view(np.uint32), self.bw[:k].cpu().numpy().view(np.uint32),
                                   self.bh[:k].cpu().numpy().view(np.uint32), self.mn[:k].cpu().numpy().view(np.uint32), self.mx[:k].cpu().numpy().view(np.uint32))

    def headline_cpu_baseline(self):
        """CPU baseline: the reference's own multithreaded reduce on host cores."""
        from oracle import pyoracle as po
        a = self.a
        cores = os.cpu_count() or 1
        kind = "reference" if po.have_ref() else "port"
        thr = cores if kind == "reference" else 1
        ct = a.cpu_tiles or max(2, min(a.tiles, 4 * thr if kind == "reference" else 8))
        k = ct * self.ROIS_PER_TILE
        hb = self.host_slice(k)
        if kind == "reference":
            tm = []
            po.ref_featurize(hb, self.mask, self.s, n_threads=thr, timing=tm)   # reduce stage only (runParallel ladder)
            sec = tm[0]
        else:
            c0 = time.perf_counter()
            po.oracle_featurize(hb, self.mask, self.s)
            sec = time.perf_counter() - c0
        self.rec["cpu_baseline"] = {
            "value": k / sec, "unit": "ROIs/s", "cores": thr, "kind": kind,
            "sample": f"{ct} tiles ({k} ROIs) of the same workload; "
                      + ("reference PixelIntensityFeatures::reduce + GLCMFeature::parallel_process_1_batch via runParallel "
                         f"with {thr} threads, reduce stage only ({sec:.2f} s)" if kind == "reference"
                         else f"single-threaded C restatement ({sec:.2f} s)"),
            "host_cpus": cores}

    def cpu_leg(self, hb_, msk, st, what, budget_rois):
        """north_star: every ROIs/s figure "next to Nyxus's own multithreaded CPU path".  The reference's classes (oracle/_ref, its
        runParallel ladder over all host cores) -- or, where that library is absent, the single-threaded C restatement -- on the
        first `budget_rois` ROIs of the leg's own batch."""
        if self.a.no_cpu_baseline:
            return None
        from oracle import pyoracle as po
        _abi = self._abi
        cores_ = os.cpu_count() or 1
        kind_ = "reference" if po.have_ref() else "port"
        thr_ = cores_ if kind_ == "reference" else 1
        k_ = int(min(hb_.n_roi, budget_rois if kind_ == "reference" else max(64, budget_rois // 64)))
        sub = _abi.HostBatch(hb_.roi_label[:k_], hb_.px_offset[:k_ + 1], hb_.x[:int(hb_.px_offset[k_])], hb_.y[:int(hb_.px_offset[k_])],
                             hb_.inten[:int(hb_.px_offset[k_])], hb_.bbox_w[:k_], hb_.bbox_h[:k_], hb_.min_inten[:k_], hb_.max_inten[:k_])
        try:
            if kind_ == "reference":
                tm_ = []
                po.ref_featurize(sub, msk, st, n_threads=thr_, timing=tm_)
                sec_ = tm_[0]
            else:
                c0_ = time.perf_counter()
                po.oracle_featurize(sub, msk, st)
                sec_ = time.perf_counter() - c0_
        except Exception as ec:
            return {"error": repr(ec)}
        return {"value": k_ / sec_, "unit": "ROIs/s", "cores": thr_, "kind": kind_, "host_cpus": cores_,
                "sample": f"the first {k_} ROIs of this leg's batch, {what}, reduce stage only ({sec_:.2f} s)"}

    def timed(self, msk, st, cbatch, n_rows, reps=8, check_rows=None, arrays=None, all_rows=False):
        """One leg: warm-up call, `reps` timed calls; the parity gate (rows `check_rows` of the table vs the oracle) on what it left.
        Returns (seconds per call, columns, gate string or None, error margins)."""
        torch, ctx = self.torch, self.ctx
        nc = ctx.n_columns(msk, st)
        o = torch.empty((n_rows, nc), dtype=torch.float64, device=self.dev)
        ctx.featurize_device_async(cbatch, msk, st, o.data_ptr(), nc)
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        for _ in range(reps):
            ctx.featurize_device_async(cbatch, msk, st, o.data_ptr(), nc)
        torch.cuda.synchronize()
        dt_ = (time.perf_counter() - c0) / reps
        ctx.sync()                               # (reads the error flag: raises on a device-side error)
        par = None
        margin = {}
        if check_rows is not None and not self.a.no_check:
            ct = torch.from_numpy(np.asarray(check_rows, np.int64)).to(self.dev)
            par = gate(o[ct].cpu().numpy(), host_rows(arrays, check_rows), msk, st, marg=margin)
            if all_rows and par == "ok":
                inv_ = self.all_rows_invariant(o, msk, st)
                if inv_ == "ok":
                    par = f"ok (rows of the last tile vs oracle; all {n_rows} rows, MIN/MAX/RANGE/MEAN/INTEGRATED_INTENSITY vs exact device reductions)"
                elif inv_ != "n/a":
                    par = f"all {n_rows} rows vs exact device reductions: {inv_}"
        return dt_, nc, par, margin

    @staticmethod
    def hbm_roofline(alg_bytes, dt):
        return {"bound": "hbm", "achieved": alg_bytes / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_bytes / dt / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": alg_bytes}

    def legs_configs(self):
        """BASELINE.json configs[1], [2], the metric workload at the reference's DEFAULT grey depth, configs[3] per GPU -- on the resident batch."""
        _abi, a, rec, s, cb, n_roi, n_px = self._abi, self.a, self.rec, self.s, self.cb, self.n_roi, self.n_px
        cpu_rows = min(n_roi, self.ROIS_PER_TILE * max(2, min(a.tiles, os.cpu_count() or 1)))     # a tile per host thread
        hb_cpu = None if a.no_cpu_baseline else self.host_slice(cpu_rows)
        s64 = _abi.default_settings(64)
        dt2, nc2, par2, mg2 = self.timed(_abi.FAM_INTENSITY, s64, cb, n_roi, check_rows=self.last_tile, arrays=self.dev_arrays, all_rows=True)
        rec["config2"] = {"value": n_roi / dt2, "unit": "ROIs/s", "ms_per_step": 1e3 * dt2, "n_columns": nc2, "parity_check": par2, "max_rel_err": mg2,
                          "roofline": self.hbm_roofline(n_px * 4 + n_roi * nc2 * 8, dt2),                    # intensities only: 4 B per ROI pixel in
                          "what": "BASELINE.json configs[1]: *ALL_INTENSITY* alone (36 columns, 64 histogram bins = the default coarse_gray_depth) on the same 1000 tiles"}
        if hb_cpu is not None:
            rec["config2"]["cpu_baseline"] = self.cpu_leg(hb_cpu, _abi.FAM_INTENSITY, s64, "PixelIntensityFeatures::reduce via runParallel", cpu_rows)
        dt3, nc3, par3, mg3 = self.timed(_abi.FAM_GLCM, s, cb, n_roi, check_rows=self.last_tile, arrays=self.dev_arrays)
        rec["config3"] = {"value": n_roi / dt3, "unit": "ROIs/s", "ms_per_step": 1e3 * dt3, "n_columns": nc3, "parity_check": par3, "max_rel_err": mg3,
                          "roofline": self.hbm_roofline(n_px * 8 + n_roi * nc3 * 8, dt3),
                          "what": "BASELINE.json configs[2]: *ALL_GLCM* alone (8 grey levels, 4 angles, d = 1; 149 columns) on the same 1000 tiles"}
        if hb_cpu is not None:
            rec["config3"]["cpu_baseline"] = self.cpu_leg(hb_cpu, _abi.FAM_GLCM, s, "GLCMFeature::parallel_process_1_batch via runParallel", cpu_rows)
        dt64, nc64, par64, mg64 = self.timed(self.mask, s64, cb, n_roi, check_rows=self.last_tile, arrays=self.dev_arrays, all_rows=True)
        rec["gray_depth_64"] = {"value": n_roi / dt64, "unit": "ROIs/s", "ms_per_step": 1e3 * dt64, "parity_check": par64, "max_rel_err": mg64,
                                "roofline": self.hbm_roofline(n_px * 8 + n_roi * nc64 * 8, dt64),
                                "what": "the metric workload at the reference's default coarse_gray_depth=64 (64 x 64 co-occurrence matrices, 64 histogram bins)"}
        if hb_cpu is not None:
            rec["gray_depth_64"]["cpu_baseline"] = self.cpu_leg(hb_cpu, self.mask, s64, "intensity + GLCM at grey depth 64 via runParallel", cpu_rows)
        self.m4 = m4 = _abi.FAM_INTENSITY | _abi.FAM_GLCM | _abi.FAM_GLRLM | _abi.FAM_GLSZM | _abi.FAM_NGTDM
        dt4, nc4, par4, mg4 = self.timed(m4, s, cb, n_roi, check_rows=self.last_tile, arrays=self.dev_arrays, all_rows=True)
        rec["config4"] = {"value": n_roi / dt4, "unit": "ROIs/s", "ms_per_step": 1e3 * dt4, "n_columns": nc4, "parity_check": par4, "max_rel_err": mg4,
                          "roofline": self.hbm_roofline(n_px * 8 + n_roi * nc4 * 8, dt4),
                          "what": "BASELINE.json configs[3] per GPU: *ALL_GLCM*+*ALL_GLRLM*+*ALL_GLSZM*+*ALL_NGTDM*+*ALL_INTENSITY* (gd 8) on the same 1000 tiles"}
        if hb_cpu is not None:
            rec["config4"]["cpu_baseline"] = self.cpu_leg(hb_cpu, m4, s, "the five families' reduce functions via runParallel, one family after the other", cpu_rows)

    def leg_gabor_metric(self):
        """The Gabor family alone on the metric ROIs, default bank."""
        _abi, rec, n_roi, side = self._abi, self.rec, self.n_roi, self.side
        try:
            dtg, ncg, parg, _ = self.timed(_abi.FAM_GABOR, self.s, self.cb, n_roi, check_rows=self.last_tile[:64], arrays=self.dev_arrays)
            flg = float(5.0 * 4.0 * side * side * 256.0) * n_roi
            rec["gabor_metric"] = {"value": n_roi / dtg, "unit": "ROIs/s", "ms_per_step": 1e3 * dtg, "ms_per_196k_rois": 1e3 * dtg * 196000.0 / n_roi, "n_columns": ncg,
                                   "parity_check": parg,
                                   "roofline": gabor_roofline(flg, dtg, np.full(n_roi, side), np.full(n_roi, side), np.full(n_roi, 4095), 4),
                                   "what": "GaborFeature alone (default bank: low-pass + 4 filters, 16 x 16 taps) on the metric workload's ROIs"}
        except Exception as eg:
            rec["gabor_metric"] = {"error": repr(eg)}

    def leg_config5(self):
        """BASELINE.json configs[4]: GABOR (8-filter bank) + ZERNIKE2D on DSB2018-shaped ROIs."""
        torch, _abi, rec = self.torch, self._abi, self.rec
        try:
            from tests import fixtures
            rng5 = np.random.default_rng(5)
            rois5 = []
            for _k in range(6000):
                for d5 in fixtures.reference_tests()["dsb2018"]:
                    r5 = fixtures.dsb_roi(d5)
                    v5 = r5["inten"].astype(np.int64)
                    v5 = np.where(v5 > 0, np.clip(v5 + rng5.integers(-8, 9, len(v5)), 1, 255), 0).astype(np.uint32)
                    m5 = v5 > 0
                    rois5.append(dict(x=r5["x"][m5], y=r5["y"][m5], inten=v5[m5]))
            hb5 = _abi.batch_from_rois(rois5)
            s5 = _abi.default_settings(64)
            s5.gabor_n_filters = 8
            for i5 in range(8):
                s5.gabor_f0[i5] = [4.0, 16.0, 32.0, 64.0][i5 % 4]
                s5.gabor_theta[i5] = np.pi * i5 / 8
            keep5 = {k5: torch.from_numpy(getattr(hb5, k5).view({2: np.int16, 4: np.int32, 8: np.int64}[getattr(hb5, k5).dtype.itemsize])).to(self.dev)
                     for k5 in ("px_offset", "x", "y", "inten", "bbox_w", "bbox_h", "min_inten", "max_inten")}
            cb5 = hb5.c_struct()
            for k5, t5 in keep5.items():
                setattr(cb5, k5, t5.data_ptr())
            cb5.slide_min = None; cb5.slide_max = None; cb5.memory = _abi.MEM_DEVICE
            m5k = _abi.FAM_GABOR | _abi.FAM_ZERNIKE
            chk5 = np.concatenate([np.arange(64), np.arange(hb5.n_roi - 64, hb5.n_roi)])       # first and last ROIs of the batch
            dt5, _, par5, mg5 = self.timed(m5k, s5, cb5, hb5.n_roi, check_rows=chk5, arrays=keep5)
            # Gabor 2*2*w*h*n^2 flops per filter (complex MAC on a real image; 8 filters + the low-pass), Zernike 2*55 per pixel
            fl5 = float(np.sum(9.0 * 4.0 * hb5.bbox_w.astype(np.float64) * hb5.bbox_h * 256.0) + 110.0 * hb5.n_px)
            rec["config5"] = {"value": hb5.n_roi / dt5, "unit": "ROIs/s", "ms_per_step": 1e3 * dt5, "rois": int(hb5.n_roi), "mean_px": hb5.n_px / hb5.n_roi,
                              "parity_check": par5, "max_rel_err": mg5,
                              "roofline": dict(gabor_roofline(fl5, dt5, hb5.bbox_w, hb5.bbox_h, hb5.max_inten, 8),
                                               note="the eight band-pass filters run on the matrix pipe (v_mfma_f32_16x16x32_f16 over f16 digit planes, two groups of four filters; "
                                                    "DESIGN 4.4), the low-pass filter as a separable packed-fp32 pass, pixels inside the error band in fp64: peak = the dense f16 MFMA "
                                                    "rate (MI355X_MICROARCH.md).  `achieved` counts SURVEY 8(d)'s 2*2*w*h*n^2 flops per filter; `executed_mfma_flops_per_launch` what "
                                                    "the stage issues (16-row tiles x 4-column groups x hi / lo tap parts x digit planes): small boxes fill a fraction of a tile"),
                              "what": "BASELINE.json configs[4]: GABOR (8-filter bank, 16x16) + ZERNIKE2D on DSB2018-shaped ROIs (fixture shapes replicated with seeded noise)"}
            rec["config5"]["cpu_baseline"] = self.cpu_leg(hb5, m5k, s5, "GaborFeature + ZernikeFeature reduce via runParallel", 8192)
            # the same DSB2018-shaped batch (real nuclei: ~130 px, most of them in the smallest size class) through the families of
            # BASELINE configs[3] and through the metric families at the reference's default grey depth
            dsb = {"rois": int(hb5.n_roi), "mean_px": hb5.n_px / hb5.n_roi,
                   "what": "the config5 batch (DSB2018 fixture shapes, 8-bit intensities) through other family sets: ns per ROI, gated on the first / last 64 ROIs"}
            for key5, mk5, st5 in (("config4_set", self.m4, self.s), ("metric_families", self.mask, self.s), ("metric_families_gd64", self.mask, _abi.default_settings(64))):
                dtd, _, pard, _ = self.timed(mk5, st5, cb5, hb5.n_roi, check_rows=chk5, arrays=keep5)
                dsb[key5] = {"value": hb5.n_roi / dtd, "unit": "ROIs/s", "ns_per_roi": 1e9 * dtd / hb5.n_roi, "parity_check": pard}
            rec["dsb_shaped"] = dsb
            del keep5
        except Exception as e5:           # informational leg: never costs the headline line
            rec["config5"] = {"error": repr(e5)}

    def legs_sizes(self):
        """ROI size: homogeneous batches at six sizes, and one heavy-tailed batch (log-normal radii 4..150 + 1 % of 300..400-px boxes)
        in ONE call -- the reference's workers take ROIs of any size (parallel.h:23-42); here a call is split into launches per size
        class (nyxhip_launch_report).  The mixed batch runs with the metric families, BASELINE configs[3]'s five, and all twelve."""
        torch, _abi, a, rec, ctx, dev, mask, s = self.torch, self._abi, self.a, self.rec, self.ctx, self.dev, self.mask, self.s
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import size_legs as sl
            rec["size_sweep"] = {"rows": sl.size_sweep(ctx, dev, mask, s),
                                 "what": "ns per ROI of homogeneous batches of disks (radius 4, 9, 18, 30, 51, 102), *ALL_GLCM*+*ALL_INTENSITY*, gd 8"}
            # the same at the reference's default grey depth, small boxes (where the 64 x 64 feature pass is the whole cost)
            rec["size_sweep_gd64"] = {"rows": sl.size_sweep(ctx, dev, mask, _abi.default_settings(64), radii=(4, 9, 18, 30)),
                                      "what": "ns per ROI of homogeneous batches of disks (radius 4, 9, 18, 30), *ALL_GLCM*+*ALL_INTENSITY*, grey depth 64"}

            def chk_mixed(bm, om, mk=mask):
                rng = np.random.default_rng(1)
                pick = np.unique(np.concatenate([rng.choice(bm.n_roi, 96, replace=False), [int(np.argmax(bm.n_px_roi))], [int(np.argmin(bm.n_px_roi))], [bm.n_roi - 1]]))
                ct = torch.from_numpy(pick).to(dev)
                return gate(om[ct].cpu().numpy(), bm.host_rows(pick), mk, s)
            # ---- intensity range: the metric ROI (disk r = 30) with 8-bit and 16-bit intensities.  The order-statistics engine
            #      follows the range: 16-bit counting table up to 16383 (the metric's 12-bit data), radix sort beyond -----------
            rng_rows = []
            for hi_ in (256, 4096, 65536):
                bb = sl.DeviceBatch([(30, 30)] * 50_000, dev, seed=11, hi=hi_)
                ob = torch.empty((bb.n_roi, self.ncol), dtype=torch.float64, device=dev)
                dtb = sl.time_call(ctx, bb, mask, s, ob)
                parb = None if a.no_check else gate(ob[:64].cpu().numpy(), bb.host_rows(np.arange(64)), mask, s)
                rng_rows.append({"intensities": f"U[1, {hi_ - 1}]", "ns_per_roi": 1e9 * dtb / bb.n_roi, "rois_per_s": bb.n_roi / dtb, "parity_check": parb})
                del bb, ob
            rec["intensity_range"] = {"rows": rng_rows, "what": "50 000 metric ROIs (disk r = 30, 2821 px), *ALL_GLCM*+*ALL_INTENSITY*, gd 8, at three intensity depths"}
            mrec, bm, om = sl.mixed_sizes(ctx, dev, mask, s, check=None if a.no_check else chk_mixed)
            del om
            keys = ("ms_per_call", "rois_per_s", "GBps", "classes", "parity_check")
            m4rec, _, _ = sl.mixed_sizes(ctx, dev, self.m4, s, check=None if a.no_check else (lambda b_, o_: chk_mixed(b_, o_, self.m4)))
            mrec["config4_set"] = {k_: m4rec[k_] for k_ in keys if k_ in m4rec}
            # every family the library has in one call on the same heavy-tailed batch (the reference hands any ROI to any worker for
            # every family, parallel.h:34-41): GLDZM / GLDM / NGLDM, contour + moments, Zernike and Gabor of the ROIs beyond LDS still run
            # one workgroup per ROI from the global workspace
            marec, _, _ = sl.mixed_sizes(ctx, dev, _abi.FAM_ALL, s, check=None if a.no_check else (lambda b_, o_: chk_mixed(b_, o_, _abi.FAM_ALL)))
            mrec["all_families"] = {k_: marec[k_] for k_ in keys if k_ in marec}
            rec["mixed_sizes"] = mrec
            del bm
        except Exception as es:
            rec["mixed_sizes"] = {"error": repr(es)}

    # ---- informational: the fused tile path (label scan + ROI assembly + reduce from tiles in HBM) -------
    def leg_tile_path(self):
        torch, _abi, _lib, a, rec, ctx, dev, mask, s, ncol = self.torch, self._abi, self._lib, self.a, self.rec, self.ctx, self.dev, self.mask, self.s, self.ncol
        from tests import synth
        nt = a.tile_path_tiles
        lab1 = torch.from_numpy(synth.disk_label_tile().astype(np.int32)).to(dev)
        labs = lab1.unsqueeze(0).repeat(nt, 1, 1).contiguous()
        tin = torch.randint(1, 4096, (nt, 1024, 1024), generator=self.gen, device=dev, dtype=torch.int32)
        cap = nt * 196
        t_lab = torch.empty(cap, dtype=torch.int32, device=dev)
        t_idx = torch.empty(cap, dtype=torch.int32, device=dev)
        t_out = torch.empty((cap, ncol), dtype=torch.float64, device=dev)
        nroi = C.c_uint64(0)
        lib = _lib.load()

        def tile_step(label_stack):
            rc = lib.nyxhip_featurize_tiles(ctx._h, tin.data_ptr(), label_stack.data_ptr(), 1024, 1024, nt, _abi.MEM_DEVICE, 196, mask,
                                            C.byref(s), t_lab.data_ptr(), t_idx.data_ptr(), cap, t_out.data_ptr(), ncol, C.byref(nroi))
            if rc != 0:
                raise RuntimeError(lib.nyxhip_last_error(ctx._h).decode())

        def tile_gate(label_stack):
            """Parity gate of a device tile-path leg: the rows of the LAST tile vs host ROI assembly + oracle, then every row of the
            call against exact device reductions of its (tile, label) pixels."""
            if a.no_check:
                return None
            n_ = int(nroi.value)
            tile_gate.margin = {}
            sel = torch.nonzero(t_idx[:n_] == nt - 1).flatten()
            from tests import roi_assembly
            hbt = roi_assembly.assemble(tin[nt - 1].cpu().numpy().view(np.uint32), label_stack[nt - 1].cpu().numpy().view(np.uint32), 1.7976931348623157e308, -1.7976931348623157e308)
            if hbt is None or hbt.n_roi != len(sel) or not np.array_equal(t_lab[:n_][sel].cpu().numpy().view(np.uint32), hbt.roi_label):
                return "ROW MISMATCH (labels of the last tile)"
            par = gate(t_out[:n_][sel].cpu().numpy(), hbt, mask, s, marg=tile_gate.margin)
            cols = _lib.column_names(mask, s)
            if par != "ok" or "MIN" not in cols:
                return par
            K = int(label_stack.max().item()) + 1
            big = torch.iinfo(torch.int64).max
            key_r = t_idx[:n_].to(torch.int64) * K + t_lab[:n_].to(torch.int64)
            wrong = {c: 0 for c in ("MIN", "MAX", "RANGE", "MEAN", "INTEGRATED_INTENSITY")}
            for t0_ in range(0, nt, 16):
                t1_ = min(nt, t0_ + 16)
                Lk = label_stack[t0_:t1_].reshape(t1_ - t0_, -1).to(torch.int64) + K * torch.arange(t1_ - t0_, device=dev, dtype=torch.int64)[:, None]
                Vk = tin[t0_:t1_].reshape(t1_ - t0_, -1).to(torch.int64)
                Lk = Lk.reshape(-1); Vk = Vk.reshape(-1)
                sz = (t1_ - t0_) * K
                sm = torch.zeros(sz, dtype=torch.int64, device=dev).scatter_add_(0, Lk, Vk)
                cn = torch.zeros(sz, dtype=torch.int64, device=dev).scatter_add_(0, Lk, torch.ones_like(Vk))
                lo_ = torch.full((sz,), big, dtype=torch.int64, device=dev).scatter_reduce_(0, Lk, Vk, "amin")
                hi_ = torch.zeros(sz, dtype=torch.int64, device=dev).scatter_reduce_(0, Lk, Vk, "amax")
                rows_ = torch.nonzero((t_idx[:n_] >= t0_) & (t_idx[:n_] < t1_)).flatten()
                kk = key_r[rows_] - t0_ * K
                tot_ = sm[kk].to(torch.float64)
                exact = {"MIN": lo_[kk].to(torch.float64), "MAX": hi_[kk].to(torch.float64), "RANGE": (hi_[kk] - lo_[kk]).to(torch.float64),
                         "INTEGRATED_INTENSITY": tot_, "MEAN": torch.div(tot_, cn[kk].to(torch.float64))}
                for c_, v_ in exact.items():
                    wrong[c_] += int((t_out[:n_][rows_, cols.index(c_)] != v_).sum().item())
                del Lk, Vk, sm, cn, lo_, hi_
            if any(wrong.values()):
                return f"all {n_} rows vs exact device reductions: MISMATCHES " + json.dumps({c: k for c, k in wrong.items() if k})
            return f"ok (rows of the last tile vs oracle; all {n_} rows, MIN/MAX/RANGE/MEAN/INTEGRATED_INTENSITY vs exact device reductions of the tiles)"
        reps = 6

        def time_stack(label_stack):
            tile_step(label_stack)
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            for _ in range(reps):
                tile_step(label_stack)
            torch.cuda.synchronize()
            return (time.perf_counter() - c0) / reps
        dt = time_stack(labs)
        par_t = tile_gate(labs)
        tile_bytes = nt * (8 * 1024 * 1024 + 196 * ncol * 8)     # 8.68 MB per tile
        # measured HBM traffic of the path's kernels per tile (profiles/hbm_traffic.json "tile_path_bytes_per_tile": scan + window-mode reduce
        # + GLCM features + the small kernels; replayed like the headline's, valid for the sources it names)
        tp_traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
            tp_traffic = tj.get("tile_path_bytes_per_tile")
        except Exception:
            pass
        rec["tile_path"] = {"value": nroi.value / dt, "unit": "ROIs/s", "tiles_per_s": nt / dt, "tiles": nt,
                            "rois": int(nroi.value), "ms_per_call": 1e3 * dt, "parity_check": par_t, "max_rel_err": getattr(tile_gate, "margin", None),
                            "algorithmic_GBps": tile_bytes / dt / 1e9, "hbm_frac": tile_bytes / dt / 1e9 / HBM_PEAK_GBS,
                            "traffic_bytes_per_tile": tp_traffic, "traffic_ratio": (tp_traffic / (tile_bytes / nt)) if tp_traffic else None,
                            "traffic_source": "replayed from profiles/hbm_traffic.json (profiles/r06f_tilepath_traffic.txt)" if tp_traffic else None,
                            "what": "nyxhip_featurize_tiles on uint32 intensity+label tiles resident in HBM: device label scan, "
                                    "compaction, label ranking, then the reduce kernels reading each ROI's bounding-box window of its tile "
                                    "(no materialised clouds; one host sync inside for the ROI count)"}
        # SURVEY 8(d)'s second synthetic set on the same path: per-ROI radius in [8, 36), 10 % concave ROIs (mixed sizes: boxes
        # on both sides of a wave's width, load imbalance, background inside the boxes); eight distinct label tiles, cycled
        try:
            nvar = 8
            lab_i = torch.from_numpy(np.stack([synth.disk_label_tile(irregular=True, seed=k) for k in range(nvar)]).astype(np.int32)).to(dev)
            labs_i = lab_i.repeat((nt + nvar - 1) // nvar, 1, 1)[:nt].contiguous()
            dti = time_stack(labs_i)
            rec["tile_path"]["irregular"] = {"value": nroi.value / dti, "unit": "ROIs/s", "tiles": nt, "rois": int(nroi.value), "ms_per_call": 1e3 * dti,
                                             "parity_check": tile_gate(labs_i),
                                             "what": "the same call on SURVEY 8(d)'s irregular label tiles (radius 8..35 per ROI, 10 % concave): "
                                                     "mixed ROI sizes, bounding boxes up to 71 wide"}
            del labs_i, lab_i
        except Exception as ei:             # informational leg: never costs the headline line
            rec["tile_path"]["irregular"] = {"error": repr(ei)}
        # CPU baseline of THIS leg: the reference's in-memory workflow end to end (its two serial label scans with a hash-map
        # lookup per pixel + ROI buffers + the multithreaded reduce) on a bounded sample of the same tiles
        if not a.no_cpu_baseline:
            from oracle import pyoracle as po
            if po.have_ref():
                cores = os.cpu_count() or 1
                thr = max(1, min(cores, 32))
                ns = min(nt, 16)
                tmc = []
                c0 = time.perf_counter()
                _, cl, _ = po.ref_featurize_tiles(tin[:ns].cpu().numpy().view(np.uint32), labs[:ns].cpu().numpy().view(np.uint32), mask, s,
                                                  n_threads=thr, timing=tmc)
                wall = time.perf_counter() - c0
                rec["tile_path"]["cpu_baseline"] = {
                    "value": len(cl) / (tmc[0] + tmc[1]), "unit": "ROIs/s", "cores": thr, "kind": "reference", "host_cpus": cores,
                    "scan_seconds": tmc[0], "reduce_seconds": tmc[1], "wall_seconds": wall,
                    "sample": f"{ns} of the same tiles ({len(cl)} ROIs): the reference's in-memory workflow per image pair -- phase-1 and phase-2 "
                              "label scans (serial, hash-map lookup per pixel: phase1.cpp:373-409, phase2_2d.cpp:637-684), ROI buffers, then "
                              f"the runParallel reduce with {thr} threads (oracle/ref_driver.cpp nyxref_featurize_tiles)"}
        # PCIe-inclusive variants (what Nyxus.featurize() pays): host tiles in, host table out, through the chunked
        # copy / compute pipeline of nyxhip_featurize_tiles_v2 -- uint32 tiles, and the same images in the element types a
        # microscope hands over (uint16 intensities, uint8 labels: H2D carries 3 B per pixel instead of 8).  Gate: the rows of the
        # LAST tile of what the last call returned vs host ROI assembly + oracle.
        nh = min(nt, 256)                        # (2 GiB of host tiles; the call itself goes through them in chunks)
        h_in32 = tin[:nh].cpu().numpy().view(np.uint32)
        h_lab32 = labs[:nh].cpu().numpy().view(np.uint32)
        def own_map(arr):
            """The same bytes in an anonymous mapping of its own: what NYXHIP_MEM_HOST_OWN_MAPPING asks the caller to state."""
            import mmap
            m = mmap.mmap(-1, max(arr.nbytes, 4096))
            out = np.frombuffer(m, dtype=arr.dtype, count=arr.size).reshape(arr.shape)
            out[...] = arr
            return out
        legs_h = (("pcie_inclusive", h_in32, h_lab32, False), ("pcie_inclusive_u16_u8", h_in32.astype(np.uint16), h_lab32.astype(np.uint8), False),
                  ("pcie_inclusive_own_mapping", own_map(h_in32), own_map(h_lab32), True))
        for tag, hi, hl, own in legs_h:
            ctx.featurize_tiles_host(hi, hl, mask, s, own_mapping=own)
            c0 = time.perf_counter()
            for _ in range(2):
                htile_out, hl_out, htab_out = ctx.featurize_tiles_host(hi, hl, mask, s, own_mapping=own)
            dth = (time.perf_counter() - c0) / 2
            par_h = None
            if not a.no_check:
                try:
                    from tests import roi_assembly
                    selh = np.nonzero(np.asarray(htile_out) == nh - 1)[0]
                    hbh = roi_assembly.assemble(h_in32[nh - 1], h_lab32[nh - 1], 1.7976931348623157e308, -1.7976931348623157e308)
                    if hbh is None or hbh.n_roi != len(selh) or not np.array_equal(np.asarray(hl_out)[selh].astype(np.uint32), hbh.roi_label):
                        par_h = "ROW MISMATCH (labels of the last tile)"
                    else:
                        par_h = gate(np.asarray(htab_out)[selh], hbh, mask, s)
                except Exception as eh:
                    par_h = "FAILED: " + repr(eh)
            rec["tile_path"][tag] = {"value": len(hl_out) / dth, "unit": "ROIs/s", "tiles": nh, "ms_per_call": 1e3 * dth,
                                     "host_GBps": (hi.nbytes + hl.nbytes) / dth / 1e9, "parity_check": par_h,
                                     "what": ("host tiles declared mappings of their own (NYXHIP_MEM_HOST_OWN_MAPPING: registered, DMA in place; " if own else
                                              "pageable host tiles through the library's pinned staging ring (NYXHIP_MEM_HOST; ")
                                             + str(hi.dtype) + " intensity, " + str(hl.dtype) + " labels) in, host table out"}
        # "(a) end-to-end featurize() (tile arrays in host memory -> feature table in host memory) and (b) reduce
        # stage only ...  Both are reported; the headline ratio is (a) vs (a)."
        cb_t = rec["tile_path"].get("cpu_baseline") or {}
        if cb_t.get("wall_seconds") and rec["tile_path"].get("pcie_inclusive", {}).get("value"):
            cpu_a = (cb_t["value"] * (cb_t["scan_seconds"] + cb_t["reduce_seconds"])) / cb_t["wall_seconds"]      # ROIs / wall second of the workflow
            rec["ratios"] = {"a_vs_a_end_to_end": rec["tile_path"]["pcie_inclusive"]["value"] / cpu_a,
                             "a_vs_a_u16_u8": rec["tile_path"]["pcie_inclusive_u16_u8"]["value"] / cpu_a,
                             "tiles_resident_vs_cpu_workflow": rec["tile_path"]["value"] / cb_t["value"],
                             "b_vs_b_reduce_stage": (self.value / rec["cpu_baseline"]["value"]) if rec.get("cpu_baseline", {}).get("value") else None,
                             "what": "GPU ROIs/s over the reference's CPU ROIs/s on this box: (a) host tiles -> host table against the reference's in-memory "
                                     "workflow (scans + reduce, wall clock); (b) the reduce stage alone (the headline metric against cpu_baseline)"}

    def summary(self):
        """The driver keeps the TAIL of the line: one number + the gate + the CPU figure + the roofline fraction per leg, last."""
        rec = self.rec

        def brief(o, key="value"):
            if not isinstance(o, dict):
                return None
            cb_ = o.get("cpu_baseline") if isinstance(o.get("cpu_baseline"), dict) else {}
            pc = o.get("parity_check")
            rf = o.get("roofline") if isinstance(o.get("roofline"), dict) else {}
            frac = rf.get("frac", o.get("hbm_frac"))
            return [o.get(key), "ok" if (pc is not None and gate_ok(pc)) else ("unchecked" if pc is None else "FAILED"), cb_.get("value"),
                    round(frac, 4) if isinstance(frac, (int, float)) else None]
        summ = {"headline": [rec.get("value"), "ok" if gate_ok(rec["config"].get("parity_check")) else "FAILED", (rec.get("cpu_baseline") or {}).get("value"),
                             round(rec["roofline"]["frac"], 4)],
                "columns": ["GPU ROIs/s (ms_per_call for mixed_sizes)", "parity gate", "CPU ROIs/s (reference classes, all host cores)",
                            "fraction of the roofline that bounds the leg (HBM 8 TB/s; config5: dense f16 MFMA)"]}
        for leg in ("config2", "config3", "config4", "config5", "gray_depth_64", "tile_path"):
            if leg in rec:
                summ[leg] = brief(rec[leg])
        if isinstance(rec.get("gabor_metric"), dict) and "ms_per_196k_rois" in rec["gabor_metric"]:
            summ["gabor_metric_ms_per_196k_rois"] = [round(rec["gabor_metric"]["ms_per_196k_rois"], 2), "ok" if gate_ok(rec["gabor_metric"].get("parity_check")) else "FAILED"]
        if isinstance(rec.get("tile_path"), dict):
            if rec["tile_path"].get("traffic_ratio") is not None:
                summ["tile_path.traffic_ratio"] = round(rec["tile_path"]["traffic_ratio"], 3)
            for sub in ("irregular", "pcie_inclusive", "pcie_inclusive_u16_u8", "pcie_inclusive_own_mapping"):
                if sub in rec["tile_path"]:
                    summ["tile_path." + sub] = brief(rec["tile_path"][sub])
        if isinstance(rec.get("mixed_sizes"), dict) and "ms_per_call" in rec["mixed_sizes"]:
            summ["mixed_sizes"] = brief(rec["mixed_sizes"], "ms_per_call")
            for sub in ("config4_set", "all_families"):
                if isinstance(rec["mixed_sizes"].get(sub), dict):
                    summ["mixed_sizes." + sub] = brief(rec["mixed_sizes"][sub], "ms_per_call")
        if isinstance(rec.get("size_sweep"), dict):
            summ["size_sweep_ns_per_roi"] = {str(r_["n_px"]): round(r_["ns_per_roi"], 2) for r_ in rec["size_sweep"].get("rows", []) if r_.get("ns_per_roi") is not None}
        if isinstance(rec.get("size_sweep_gd64"), dict):
            summ["size_sweep_gd64_ns_per_roi"] = {str(r_["n_px"]): round(r_["ns_per_roi"], 2) for r_ in rec["size_sweep_gd64"].get("rows", []) if r_.get("ns_per_roi") is not None}
        if isinstance(rec.get("dsb_shaped"), dict):
            summ["dsb_shaped_ns_per_roi"] = {k_: [round(v_["ns_per_roi"], 2) if v_["ns_per_roi"] is not None else None, "ok" if gate_ok(v_.get("parity_check")) else ("unchecked" if v_.get("parity_check") is None else "FAILED")]
                                             for k_, v_ in rec["dsb_shaped"].items() if isinstance(v_, dict) and "ns_per_roi" in v_}
        if isinstance(rec.get("intensity_range"), dict):
            summ["intensity_range_ns_per_roi"] = {r_["intensities"]: (round(r_["ns_per_roi"], 2) if r_.get("ns_per_roi") is not None else None) for r_ in rec["intensity_range"].get("rows", [])}
        if "ratios" in rec:
            summ["ratios"] = {k_: v_ for k_, v_ in rec["ratios"].items() if k_ != "what"}
        summ["roofline_frac"] = rec["roofline"]["frac"]
        rec["summary"] = summ


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch(a))                      # nothing GPU-related has been imported at this point
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.stub:
        return run_stub(a, world, rank)
    b = Bench(a, world, rank, local_rank)
    b.make_batch()
    b.timed_steps()
    b.ranks_and_gather()
    gate_rc = 0
    if rank == 0:
        b.headline()
        if not a.no_check:
            b.headline_gate()
        if world == 1 and not a.no_cpu_baseline:
            b.headline_cpu_baseline()
        # ---- informational legs on the same resident batch: the reference's DEFAULT grey depth, BASELINE.json configs[1]-[4], ROI sizes ----
        if world == 1 and not a.no_extras and b.mask == 3:
            b.legs_configs()
            b.leg_gabor_metric()
            b.leg_config5()
            b.legs_sizes()
        if world == 1 and a.tile_path_tiles > 0:
            b.leg_tile_path()
        gate_rc = apply_gates(b.rec)         # a leg whose features do not match loses its value; exit code 4 below
        b.summary()
        print(json.dumps(b.rec))
    if world > 1:
        torch, dist = b.torch, b.dist
        grc = torch.tensor([gate_rc], dtype=torch.int32, device=b.dev)
        dist.broadcast(grc, src=0)
        gate_rc = int(grc.item())
        dist.barrier()
        dist.destroy_process_group()
    b.ctx.close()
    if b.same_device or b.gather_err:
        sys.exit(3)                          # the line is printed, but a job whose ranks shared a device (or whose gather failed) did not measure N GPUs
    if gate_rc:
        sys.exit(gate_rc)                    # a timed configuration counts only if its features match


if __name__ == "__main__":
    main()
