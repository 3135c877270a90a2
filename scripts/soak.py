#!/usr/bin/env python3
"""Soak: 60 x (MlpEnsemble.fit + 20 generate_action calls) on one pair of handles; device memory (hipMemGetInfo), torch's allocator
and the host RSS must stay flat — the trainer keeps every epoch's permutation tensor alive until the fit's final synchronize,
the planner handle cache re-stages weights after every fit.  usage: python scripts/soak.py"""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_simba_api import make_agent_parts
env, model, pol = make_agent_parts('safe_cem_mpc', seed=1)
rng = np.random.default_rng(0)
n = 3000
obs = rng.normal(0, 0.3, (n, 60)).astype(np.float32); act = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
A = rng.normal(0, 0.02, (62, 60)).astype(np.float32)
nxt = obs + np.concatenate([obs, act], 1) @ A
x = np.concatenate([obs, act], 1)
model.model.training_steps = 300
def rss():
    import psutil; return psutil.Process().memory_info().rss / 2**20
free0 = None
for it in range(60):
    model.fit(x, nxt)
    for i in range(20): pol.generate_action(obs[i])
    if it % 10 == 0:
        torch.cuda.synchronize(); gc.collect()
        free, total = torch.cuda.mem_get_info()
        used = (total - free) / 2**20
        if free0 is None: free0 = used
        print('iter %3d: device used %.0f MiB (%+.0f), torch allocated %.0f MiB, host rss %.0f MiB' % (it, used, used - free0, torch.cuda.memory_allocated() / 2**20, rss()), flush=True)
