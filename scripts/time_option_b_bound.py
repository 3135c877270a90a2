#!/usr/bin/env python3
"""SURVEY 8e Option B, bounded before building it: what could an all-reduce of elite sufficient statistics save a
rank of a candidate-sharded plan, and what would its second collective cost?

Today a rank all-gathers the N/G local scores and runs the whole select over all N candidates (replicated).  Under Option B it would
still all-gather the scores and find the k-th key over all N (the histogram passes), but compact and take moments only over ITS OWN
elites (k/G on average), then exchange < 1 KB of partial sums and finish.  Measured here on one GPU, one process playing rank 0 of 8 at
B5 (N = 65536, 8192 per rank) and at the weak-scaled B2 (N = 16000, 2000 per rank), foreign score shards stood in by frames shaped like the
rank's own (synthetic.rehearsal_score_frames):

  select_replicated_us   the select as shipped: k elites of N
  select_own_share_us    the same launch with n_elite = k/8: the histogram / count / barrier phases unchanged, compaction output and both
                         moment passes 8x smaller — a LOWER bound on Option B's select (it adds a finish kernel and packs records)
  sampler_all_us / sampler_own_us   cem_sample_kernel over all N candidates (today) vs over N/8 (Option B samples only its own)
  extra_collective_us    one more 1-KB ncclAllGather per iteration on the planner's stream through the REAL librccl with a one-rank
                         communicator (the only RCCL a one-GPU box runs): launch + kernel, no wire — a LOWER bound on an 8-rank exchange

Option B pays only if  (select_replicated - select_own_share) + (sampler_all - sampler_own)  >  extra_collective + finish kernel."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from ethz_safe_learning_amd import CemPlanner, PlannerConfig, synthetic

pb = synthetic.problem(60, 2, 5)
I, H, W = 5, 30, 8


def planner(N, k, world):
    cfg = PlannerConfig(obs_dim=60, act_dim=2, ensemble_size=5, particles=5, n_samples=N, horizon=H, n_elite=k, iterations=I,
                        scorer=pb['scorer'], act_low=pb['low'], act_high=pb['high'], noise_stddev=1e-3, world_size=world, rank=0, use_graph=False)
    pl = CemPlanner(cfg)
    pl.set_weights(pb['weights'])
    pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
    return pl


def stepwise_times(pl, frames, reps=6):
    """median us of the select and of the sampler launch per iteration (HIP events inside the library)."""
    nloc = pl.n_local
    pl.set_timing(True)
    sel, samp, n = 0.0, 0.0, 0
    for rep in range(reps):
        pl.plan_begin(pb['state'], seed=12, call=rep)
        for it in range(I):
            pl.plan_rollout(it)
            with torch.cuda.stream(pl.stream):
                pl.scores_global(sync=False)[nloc:].copy_(frames[it, nloc:])
                pl.scores_global(sync=False)[:nloc].copy_(pl.scores_local(sync=False))
            pl.plan_select(it)
        pl.plan_end()
        if rep >= 1:
            tm = pl.last_timing()
            sel += tm['select_ms']; samp += tm['sampler_ms']; n += tm['rollout_launches']
    pl.set_timing(False)
    return 1e3 * sel / n, 1e3 * samp / n


def extra_collective_us():
    """a 1-KB all-gather on a side stream of a one-rank RCCL communicator (torch.distributed 'nccl' = RCCL), back to back."""
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    src = torch.zeros(256, dtype=torch.float32, device='cuda')
    dst = torch.zeros(256, dtype=torch.float32, device='cuda')
    for _ in range(20):
        dist.all_gather_into_tensor(dst, src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for _ in range(n):
        dist.all_gather_into_tensor(dst, src)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / n
    dist.destroy_process_group()
    return us


for name, N, k in (('B5 rank of 8', 65536, 6554), ('weak-scaled B2 rank of 8', 16000, 1600)):
    pl = planner(N, k, W)
    frames = synthetic.rehearsal_score_frames(pl, pb['state'], W, I, seed=11)
    sel_full, samp_all = stepwise_times(pl, frames)
    mode = pl.select_mode()
    pl.close()
    pl = planner(N, max(1, k // W), W)
    sel_own, _ = stepwise_times(pl, frames)
    pl.close()
    pl = planner(N // W, max(1, k // W), 1)             # a population of the rank's own size: what sampling only its own candidates costs
    os.environ['CEM_FORCE_SAMPLER'] = 'kernel'
    pl2 = planner(N // W, max(1, k // W), 1)
    del os.environ['CEM_FORCE_SAMPLER']
    fr1 = torch.zeros(I, N // W, device='cuda')
    pl.close()
    pl2.set_timing(True)
    samp_own, n = 0.0, 0
    for rep in range(4):
        pl2.plan(pb['state'], seed=3, call=rep)
        if rep >= 1:
            tm = pl2.last_timing(); samp_own += tm['sampler_ms']; n += tm['rollout_launches']
    pl2.close()
    print(json.dumps(dict(case=name, N=N, k=k, select_mode=mode, select_replicated_us=round(sel_full, 1), select_own_share_us=round(sel_own, 1),
                          sampler_all_us=round(samp_all, 1), sampler_own_us=round(1e3 * samp_own / max(n, 1), 1))), flush=True)
print(json.dumps(dict(extra_collective_us_one_rank_rccl_1KB_allgather=round(extra_collective_us(), 1))), flush=True)
