#!/usr/bin/env python3
"""Diagnostic: where MlpEnsemble.fit's wall time goes on the host (cProfile) next to the device time of its kernels."""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_simba_api import make_agent_parts
np.random.seed(0)
env, model, pol = make_agent_parts('safe_cem_mpc', seed=1)
rng = np.random.default_rng(0)
n = 30000
obs = rng.normal(0, 0.3, (n, 60)).astype(np.float32)
act = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
A = rng.normal(0, 0.02, (62, 60)).astype(np.float32)
nxt = obs + np.concatenate([obs, act], 1) @ A + 0.002 * rng.normal(0, 1, (n, 60)).astype(np.float32)
x = np.concatenate([obs, act], 1)
model.fit(x[:4000], nxt[:4000])
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter(); pr.enable(); model.fit(x, nxt); pr.disable(); dt = time.perf_counter() - t0
print('fit wall %.3f s' % dt)
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
