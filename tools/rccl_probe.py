"""RCCL sanity on one GPU: process group of size 1, async gather of a float64 table (the collective bench.py uses at N > 1)."""
import os
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
dist.destroy_process_group()
