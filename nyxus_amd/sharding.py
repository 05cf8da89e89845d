"""Multi-GPU plumbing: one process per GPU, tiles sharded statically, and the one
exchange step the path has -- gathering each rank's feature-table block on rank 0.

The reference has no counterpart (its only parallelism is the thread fan-out of
/root/reference/src/nyx/parallel.h:23-42); rows keep the reference's order: tiles in
input order, labels ascending (src/nyx/output_2_buffer.cpp:305-306), i.e. rank blocks are
concatenated in rank order because tile ranges are contiguous per rank.
Works with any torch.distributed backend ("nccl" = RCCL over xGMI on the GPU box, "gloo"
in the CPU tests).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition [lo, hi) of n_items over `world` ranks; the first
    n_items % world ranks get one extra item."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


class TableGather:
    """Gathers [rows_r x n_cols] float64 blocks (rows_r may differ per rank) on `dst`.

    start() enqueues the collective (async) so that the next batch's kernels overlap it;
    finish() waits and returns the concatenated table on dst (None elsewhere).

    Ordering: the collective runs on torch's stream, the feature kernels on the producing context's stream (its own
    non-blocking stream unless Context.set_stream() pointed it at torch's).  Pass the context as `producer`: start() then
    waits for its kernels (Context.sync()) before the table is handed to the collective.  Without it the caller must have
    synchronised already.  A table that is still being gathered must not be overwritten by the next batch: alternate
    between two output tables (bench.py does)."""

    def __init__(self, n_cols: int, dst: int = 0, group=None):
        self.n_cols = n_cols
        self.dst = dst
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._work = None
        self._bufs: Optional[List[torch.Tensor]] = None
        self._rows: Optional[List[int]] = None
        self._local = None

    def start(self, local: torch.Tensor, rows_per_rank: Optional[List[int]] = None, producer=None):
        assert local.dim() == 2 and local.shape[1] == self.n_cols and local.dtype == torch.float64
        if producer is not None:
            producer.sync()               # the table is complete (and the device error word checked) before it is sent
        if self.world == 1:
            self._local = local
            return
        if rows_per_rank is None:  # exchange the row counts first (tiny)
            cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
            allc = [torch.zeros_like(cnt) for _ in range(self.world)]
            dist.all_gather(allc, cnt, group=self.group)
            rows_per_rank = [int(c.item()) for c in allc]
        self._rows = rows_per_rank
        mx = max(rows_per_rank)
        send = local
        if local.shape[0] != mx:  # pad to the common block size
            send = torch.zeros((mx, self.n_cols), dtype=local.dtype, device=local.device)
            send[: local.shape[0]] = local
        if self.rank == self.dst:
            if self._bufs is None or self._bufs[0].shape[0] != mx or self._bufs[0].device != local.device:
                self._bufs = [torch.empty((mx, self.n_cols), dtype=local.dtype, device=local.device)
                              for _ in range(self.world)]
            self._work = dist.gather(send.contiguous(), self._bufs, dst=self.dst, group=self.group, async_op=True)
        else:
            self._work = dist.gather(send.contiguous(), None, dst=self.dst, group=self.group, async_op=True)

    def finish(self) -> Optional[torch.Tensor]:
        if self.world == 1:
            t, self._local = self._local, None
            return t
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self.rank != self.dst:
            return None
        return torch.cat([b[:r] for b, r in zip(self._bufs, self._rows)], dim=0)


def gather_rows(local_keys: torch.Tensor, local_table: torch.Tensor, dst: int = 0, group=None, producer=None):
    """The exchange step of a sharded job: every rank holds the rows of its contiguous image range -- `local_keys`
    int64 [rows, 2] = (image index, ROI label), `local_table` float64 [rows, n_cols] -- and rank `dst` receives all of them in
    rank order, i.e. images in input order and labels ascending inside an image, the row order of the reference's table
    (/root/reference/src/nyx/output_2_buffer.cpp:305-306).  Returns (keys, table) on dst, (None, None) elsewhere."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        if producer is not None:
            producer.sync()
        return local_keys, local_table
    cnt = torch.tensor([local_table.shape[0]], dtype=torch.int64, device=local_table.device)
    allc = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(allc, cnt, group=group)
    rows = [int(c.item()) for c in allc]
    g = TableGather(local_table.shape[1], dst=dst, group=group)
    g.start(local_table, rows_per_rank=rows, producer=producer)
    table = g.finish()
    gk = TableGather(2, dst=dst, group=group)
    gk.start(local_keys.to(torch.float64), rows_per_rank=rows)      # labels < 2^32 and image indices are exact in float64
    keys = gk.finish()
    return (keys.to(torch.int64), table) if table is not None else (None, None)
