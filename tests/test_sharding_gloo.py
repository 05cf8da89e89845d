"""world_size-2 gloo test of the N>1 path: shard ranges + feature-table gather."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nyxus_amd.sharding import TableGather, shard_range


def test_shard_range_partitions_everything():
    for n in (0, 1, 7, 8, 1000, 1001):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_tiles, rois_per_tile, ncol = 5, 3, 4
        lo, hi = shard_range(n_tiles, rank, world)
        # row value encodes (tile, roi) so that order can be verified on rank 0
        rows = [[t * 100 + r + c * 0.001 for c in range(ncol)] for t in range(lo, hi) for r in range(rois_per_tile)]
        local = torch.tensor(rows, dtype=torch.float64).reshape(-1, ncol)
        g = TableGather(ncol, dst=0)
        for _ in range(2):  # reuse across steps
            g.start(local)
            full = g.finish()
        if rank == 0:
            q.put(full.numpy())
    finally:
        dist.destroy_process_group()


def test_table_gather_gloo_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    full = q.get(timeout=120)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert full.shape == (15, 4)
    want = np.array([t * 100 + r for t in range(5) for r in range(3)], dtype=np.float64)
    assert np.array_equal(np.floor(full[:, 0]), want)  # tiles in input order, ranks concatenated
