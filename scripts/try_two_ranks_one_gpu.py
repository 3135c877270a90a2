#!/usr/bin/env python3
"""Experiment: can two RCCL ranks share ONE GPU (so that the library's multi-rank path could be exercised on a one-GPU box)?
torchrun --nproc-per-node 2 scripts/try_two_ranks_one_gpu.py   -> prints what RCCL says."""
import os, sys
import torch, torch.distributed as dist
rank = int(os.environ['RANK']); world = int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
try:
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    x = torch.full((4,), float(rank), device='cuda:0')
    out = [torch.empty(4, device='cuda:0') for _ in range(world)]
    dist.all_gather(out, x)
    torch.cuda.synchronize()
    print('rank %d: all_gather over %d ranks on one GPU worked: %s' % (rank, world, [o[0].item() for o in out]), flush=True)
    dist.destroy_process_group()
except Exception as e:
    print('rank %d: RCCL refused: %s' % (rank, str(e)[:600]), flush=True)
    sys.exit(3)
