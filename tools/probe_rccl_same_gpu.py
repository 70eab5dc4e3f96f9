"""Probe: can two RCCL ranks share one GPU on this box? (decides how the C-level communicator is tested)"""
import os, torch, torch.distributed as dist
dist.init_process_group("nccl")
r = dist.get_rank()
torch.cuda.set_device(0)
x = torch.full((4,), float(r + 1), device="cuda", dtype=torch.float64)
dist.all_reduce(x)
torch.cuda.synchronize()
print("rank", r, "allreduce ->", x.tolist(), flush=True)
dist.destroy_process_group()
