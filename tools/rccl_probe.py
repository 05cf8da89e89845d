"""RCCL sanity on one GPU: process group of size 1, async gather of a float64 table (the collective bench.py uses at N > 1)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.arange(12, dtype=torch.float64, device=dev).reshape(3, 4)
bufs = [torch.empty_like(x)]
w = dist.gather(x, bufs, dst=0, async_op=True)
w.wait()
torch.cuda.synchronize()
print("gather ok:", torch.equal(bufs[0], x))
t = torch.ones(1, device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
print("all_reduce/barrier ok")
# the bench's own objects on the initialised group: TableGather / gather_rows (world 1: the collective is skipped by design, the
# producer sync and the row bookkeeping run), then a raw dist.gather of one step's table (196 000 x 185 float64 = 290 MB) through RCCL
import time
from nyxus_amd.sharding import TableGather, gather_rows
tab = torch.rand((196000, 185), dtype=torch.float64, device=dev)
g = TableGather(185, dst=0)
g.start(tab, rows_per_rank=[196000])
full = g.finish()
print("TableGather (world 1) ok:", full is tab or torch.equal(full, tab))
keys = torch.stack([torch.zeros(196000, dtype=torch.int64, device=dev), torch.arange(196000, dtype=torch.int64, device=dev)], dim=1)
k2, t2 = gather_rows(keys, tab)
print("gather_rows (world 1) ok:", torch.equal(k2, keys) and torch.equal(t2, tab))
buf = [torch.empty_like(tab)]
torch.cuda.synchronize()
c0 = time.perf_counter()
dist.gather(tab, buf, dst=0, async_op=True).wait()
torch.cuda.synchronize()
dt = time.perf_counter() - c0
print(f"dist.gather of one step's table over RCCL (world 1, device-local): {1e3 * dt:.2f} ms, {tab.numel() * 8 / dt / 1e9:.1f} GB/s, equal: {torch.equal(buf[0], tab)}")
print("backend:", dist.get_backend(), "| torch", torch.__version__, "| HSA_ENABLE_IPC_MODE_LEGACY =", os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))
dist.destroy_process_group()
