#!/usr/bin/env python3
"""What ONE rank of a candidate-sharded plan pays, measured on one GPU (no collective: the all-gather is replaced by a
two device copies per iteration: this rank's shard and stand-ins for the other ranks' shards shaped like its own —
synthetic.rehearsal_score_frames).

  B5       N = 65536 over 8 ranks (8192 candidates per rank), K = P = E = 5, H = 30, k = 6554   (BASELINE.json configs[4])
  weak8    N = 16000 over 8 ranks (2000 per rank): bench.py --gpus 8
  single   the same per-rank candidate count as ONE rank (no replicated work beyond its own): the scaling reference

Per iteration: sample (all N sequences, replicated) / rollout / reduce / select (replicated, over all N scores), timed with
events on the planner's stream around the stepwise C-ABI calls; run it under `rocprofv3 --kernel-trace --stats` for the
per-kernel split.  Prints one JSON line per case."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ethz_safe_learning_amd import CemPlanner, PlannerConfig, synthetic

pb = synthetic.problem(60, 2, 5)
I, H = 5, 30
reps = int(os.environ.get('REPS', '6'))


def run(name, N, world, rank=0):
    cfg = PlannerConfig(obs_dim=60, act_dim=2, ensemble_size=5, particles=5, n_samples=N, horizon=H, n_elite=max(1, round(N / 10)),
                        iterations=I, scorer=pb['scorer'], act_low=pb['low'], act_high=pb['high'], noise_stddev=1e-3,
                        world_size=world, rank=rank, use_graph=False)
    pl = CemPlanner(cfg)
    pl.set_weights(pb['weights'])
    pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
    st = pl.stream
    nloc = N // world
    frames = synthetic.rehearsal_score_frames(pl, pb['state'], world, I, seed=11) if world > 1 else None   # the other ranks' shards, shaped like this one's
    torch.cuda.synchronize()
    t_roll, t_sel, t_plan = [], [], []
    for rep in range(reps + 2):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * I + 2)]
        pl.plan_begin(pb['state'], seed=11, call=rep)
        ev[0].record(st)
        for it in range(I):
            pl.plan_rollout(it)
            ev[1 + 2 * it].record(st)
            if world > 1:
                with torch.cuda.stream(st):
                    pl.scores_global(sync=False)[nloc:].copy_(frames[it, nloc:])
                    pl.scores_global(sync=False)[:nloc].copy_(pl.scores_local(sync=False))
            pl.plan_select(it)
            ev[2 + 2 * it].record(st)
        pl.plan_end()
        torch.cuda.synchronize()
        if rep >= 2:
            for it in range(I):
                t_roll.append(ev[2 * it].elapsed_time(ev[1 + 2 * it]))
                t_sel.append(ev[1 + 2 * it].elapsed_time(ev[2 + 2 * it]))
            t_plan.append(ev[0].elapsed_time(ev[2 * I]))
    pl.set_timing(True)
    k_roll, k_sel, n = 0.0, 0.0, 0
    for rep in range(3):
        pl.plan_begin(pb['state'], seed=12, call=rep)
        for it in range(I):
            pl.plan_rollout(it)
            if world > 1:
                with torch.cuda.stream(st):
                    pl.scores_global(sync=False)[nloc:].copy_(frames[it, nloc:])
                    pl.scores_global(sync=False)[:nloc].copy_(pl.scores_local(sync=False))
            pl.plan_select(it)
        pl.plan_end()
        tm = pl.last_timing()
        k_roll += tm['rollout_ms']; k_sel += tm['select_ms']; n += tm['rollout_launches']
    out = dict(case=name, N=N, world=world, candidates_per_rank=N // world, k=cfg.n_elite,
               sample_rollout_reduce_ms_per_iter=float(np.median(t_roll)), select_phase_ms_per_iter=float(np.median(t_sel)),
               plan_ms=float(np.median(t_plan)), rollout_kernel_ms=k_roll / n, select_kernel_ms=k_sel / n,
               chunks_per_tile=pl.tiles()[0], workgroups=int(len(pl.tiles()[1])))
    print(json.dumps(out), flush=True)
    pl.close()
    return out


cases = os.environ.get('CASES', 'b5,b5single,weak8,weak8single').split(',')
res = {}
if 'b5' in cases:
    res['b5'] = run('B5 rank 0 of 8', 65536, 8)
if 'b5single' in cases:
    res['b5single'] = run('B5 shard as a single rank (N=8192)', 8192, 1)
if 'weak8' in cases:
    res['weak8'] = run('weak-scaled bench rank 0 of 8 (N=16000)', 16000, 8)
if 'weak8single' in cases:
    res['weak8single'] = run('B2 single rank (N=2000)', 2000, 1)
for a, b in (('b5', 'b5single'), ('weak8', 'weak8single')):
    if a in res and b in res:
        print(json.dumps(dict(projection=a, eager_plan_ms_rank_of_8=res[a]['plan_ms'], eager_plan_ms_single=res[b]['plan_ms'],
                              efficiency_excluding_collective=res[b]['plan_ms'] / res[a]['plan_ms'])), flush=True)
