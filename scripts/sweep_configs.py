#!/usr/bin/env python3
"""Rollout-kernel time and roofline fraction at the BASELINE configs (B1..B4; B5's per-GPU shard = 8192 candidates).
The bench line is B2 only (bench.py); this is the table in DESIGN.md.  usage: python scripts/sweep_configs.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ethz_safe_learning_amd import CemPlanner, PlannerConfig, synthetic

CHUNKS = int(os.environ.get('CEM_SWEEP_CHUNKS', '0'))      # 0 = the library's choice; 1..4 forces rows/16 per tile (diagnostic)
ONLY = os.environ.get('CEM_SWEEP_ONLY', '')
NOGRAPH = bool(os.environ.get('CEM_SWEEP_NOGRAPH'))          # counter passes: every launch its own dispatch, and only a few plans

CFG = [('B1', 60, 2, 5, 500, 25), ('B2', 60, 2, 5, 2000, 30), ('B3', 60, 2, 16, 8192, 30), ('B4', 100, 12, 8, 4096, 50),
       ('B5/8 (one rank of 8)', 60, 2, 5, 8192, 30)]
out = []
for name, O, A, K, N, H in CFG:
    if ONLY and not any(name.startswith(x) for x in ONLY.split(',')):
        continue
    pb = synthetic.problem(O, A, K)
    cfg = PlannerConfig(obs_dim=O, act_dim=A, ensemble_size=K, particles=K, n_samples=N, horizon=H, n_elite=N // 10, iterations=5,
                        scorer=pb['scorer'], act_low=pb['low'], act_high=pb['high'], noise_stddev=1e-3, use_graph=not NOGRAPH, chunks_per_tile=CHUNKS,
                        precision=os.environ.get('CEM_SWEEP_PRECISION', 'fp32'))
    pl = CemPlanner(cfg); pl.set_weights(pb['weights']); pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
    for i in range(1 if NOGRAPH else (12 if N <= 2000 else 3)):      # small plans: the first ~10 run at ramping clocks (2.6 -> 2.16 ms at B2)
        pl.plan(pb['state'], seed=1, call=i)
    n = 2 if NOGRAPH else (20 if N <= 2000 else 5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        pl.plan(pb['state'], seed=1, call=10 + i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    pl.set_timing(True)
    ms, ln = 0.0, 0
    for i in range(3):
        pl.plan(pb['state'], seed=2, call=i); tm = pl.last_timing(); ms += tm['rollout_ms']; ln += tm['rollout_launches']
    pl.set_timing(False)
    fl = synthetic.flops_per_row_step(O, A) * K * N * H
    rc, tiles = pl.tiles()
    r = dict(config=name, obs=O, act=A, K=K, N=N, H=H, rows=K * N, chunks_per_tile=rc, workgroups=len(tiles), plan_ms=dt * 1e3,
             plans_per_s=1 / dt, cand_steps_per_s=5 * N * H / dt, rollout_ms=ms / ln, tflops=fl / (ms / ln * 1e-3) / 1e12,
             frac_of_157_3=fl / (ms / ln * 1e-3) / 157.3e12)
    out.append(r)
    print(json.dumps(r), flush=True)
    pl.close(); del pl
