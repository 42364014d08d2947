#!/usr/bin/env python
"""RCCL sanity on a 1-GPU box: backend 'nccl' with world_size 1 -- process-group init with device_id,
all_reduce of a 4.8 MB f32 buffer on a side stream (what NerfppTrainer does per level), barrier."""
import os
import time
import torch
import torch.distributed as dist

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
g = torch.ones(1202440, device=dev)
side = torch.cuda.Stream(device=dev)
for it in range(3):
    ev = torch.cuda.Event(); ev.record(); side.wait_event(ev)
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        dist.all_reduce(g)
        done = torch.cuda.Event(); done.record()
    torch.cuda.current_stream().wait_event(done)
    torch.cuda.synchronize()
    print('all_reduce #%d ok: %.3f ms, sum=%.1f' % (it, (time.perf_counter() - t0) * 1e3, float(g.sum())))
dist.barrier()
t = torch.tensor([1.5], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
print('barrier + MAX all_reduce ok', float(t))
dist.destroy_process_group()
