"""RCCL API smoke on one GPU (world size 1): the calls the data-parallel path makes -- init, all_reduce(AVG) on a flat
buffer, broadcast, barrier, all_reduce(MAX) on a float64 scalar (bench.py)."""
import os
import torch
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
flat = torch.arange(11_381_315, device='cuda', dtype=torch.float32)
dist.all_reduce(flat, op=dist.ReduceOp.AVG)
dist.broadcast(flat, 0)
dist.barrier()
t = torch.tensor([1.5], device='cuda', dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
print('ok', float(flat[-1]), float(t), dist.get_backend())
dist.destroy_process_group()
