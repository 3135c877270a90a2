#!/usr/bin/env python3
"""Diagnostic: a B2-shaped plan (obs 60, act 2, K = 5, N = 2000, H = 30, I = 5) at hidden widths beyond the fast kernel's 128
units (cem_rollout_wide_kernel), and the 15-member training step at the same widths (GEMM-by-GEMM kernel, row stride 256).
usage: python scripts/time_wide_units.py [units ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests import helpers as hp
from ethz_safe_learning_amd.trainer import CemTrainer

for units in [int(a) for a in sys.argv[1:]] or [128, 160, 256]:
    act = os.environ.get('CEM_WIDE_ACT', 'relu')       # anything but relu runs the generic kernel at any width
    pb = hp.make_problem(60, 2, 5, 4, seed=1, units=units, activation=act)
    _, cfg = hp.configs(pb, N=2000, H=30, P=5, E=5, k=200, I=5, use_graph=True)
    pl = hp.make_planner(pb, cfg)
    for i in range(12):
        pl.plan(pb['state'], seed=1, call=i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20):
        pl.plan(pb['state'], seed=1, call=20 + i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    fl = 2 * (62 * units + 3 * units * units + 2 * units * 60) * 5 * 2000 * 30 * 5
    pl.close()
    if os.environ.get('CEM_WIDE_PLAN_ONLY'):
        print('units %3d %s: plan %.3f ms (%.1f TFLOP/s of its algorithmic FLOPs)' % (units, act, dt * 1e3, fl / dt / 1e12), flush=True)
        continue
    tr = CemTrainer(62, 60, units, 4, 15)
    tr.set_state(hp.make_problem(60, 2, 15, 4, seed=1, units=units)['weights'])
    rng = np.random.default_rng(0); n = 4096
    x = torch.from_numpy(rng.standard_normal((n, 62)).astype(np.float32)).cuda()
    y = torch.from_numpy((0.1 * rng.standard_normal((n, 60))).astype(np.float32)).cuda()
    perm = torch.from_numpy(np.stack([rng.permutation(n) for _ in range(15)]).astype(np.int32)).cuda()
    loss = torch.zeros((200, 15), device='cuda')
    for i in range(5):
        tr.step(x, y, perm, 64 * i, 64, 2.5e-4, loss[i])
    tr.synchronize(); t0 = time.perf_counter()
    for i in range(200):
        tr.step(x, y, perm, (64 * i) % 4000, 64, 2.5e-4, loss[i])
    tr.synchronize(); ts = (time.perf_counter() - t0) / 200
    tr.close()
    print('units %3d: plan %.2f ms (%.1f TFLOP/s of its algorithmic FLOPs), training step %.1f us' % (units, dt * 1e3, fl / dt / 1e12, ts * 1e6), flush=True)
