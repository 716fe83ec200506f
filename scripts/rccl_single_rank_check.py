#!/usr/bin/env python3
"""The collectives bench.py issues at N > 1 (gather of the result records to rank 0 on a side stream, all_reduce of the timing pair,
barrier), on the nccl (= RCCL) backend with a one-rank group on one GPU: checks that this PyTorch / RCCL build accepts exactly these
calls on device tensors and that the gathered row is the record. Not a scaling measurement; the pool offers one GPU per box."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import importlib

sharding = importlib.import_module("srslte-emane_amd.sharding")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)  # as bench.py does
res = torch.arange(1000, dtype=torch.int64, device=dev).to(torch.uint8)
gath = torch.zeros((1, 1000), dtype=torch.uint8, device=dev)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    rows = [gath[0]]
    dist.gather(res, rows, dst=0)  # the call sharding.gather_results makes when world > 1
    host = torch.zeros((1, 1000), dtype=torch.uint8).pin_memory()
    host.copy_(gath, non_blocking=True)
side.synchronize()
assert torch.equal(host[0], res.cpu())
t = torch.tensor([1.5, 1.0], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
print("rccl one-rank check ok:", t.tolist(), sharding.reduce_counts([3, 4], dist, dev))
dist.destroy_process_group()
