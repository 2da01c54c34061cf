"""Probe (GPU box): can two ranks of a torch.distributed job over RCCL share ONE MI355X?  Run as
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 tools/probe_two_ranks_one_gpu.py"""
import os
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
x = torch.full((1 << 20,), float(rank + 1), device=dev)
dist.all_reduce(x)
torch.cuda.synchronize()
print(f"rank {rank}: all_reduce over {world} ranks on one GPU -> {float(x[0])} (expected {world * (world + 1) / 2})", flush=True)
dist.barrier()
dist.destroy_process_group()
