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
    finish() waits and returns the concatenated table on dst (None elsewhere)."""

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

    def start(self, local: torch.Tensor, rows_per_rank: Optional[List[int]] = None):
        assert local.dim() == 2 and local.shape[1] == self.n_cols and local.dtype == torch.float64
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
