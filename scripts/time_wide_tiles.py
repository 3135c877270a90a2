#!/usr/bin/env python3
"""Diagnostic: the width-generic rollout kernel at tile counts around the chip's resident slots (256 CUs x 2 workgroups):
a B2-shaped plan (obs 60, act 2, K = 5, H = 30, I = 5) at N chosen for 1, 2, 2.44 (B2) and 4 tiles per CU.
usage: python scripts/time_wide_tiles.py [units]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import helpers as hp

units = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pb = hp.make_problem(60, 2, 5, 4, seed=1, units=units)
for N in (800, 1600, 2000, 2400, 3200):
    _, cfg = hp.configs(pb, N=N, H=30, P=5, E=5, k=N // 10, I=5, use_graph=True)
    pl = hp.make_planner(pb, cfg)
    for i in range(6):
        pl.plan(pb['state'], seed=1, call=i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(12):
        pl.plan(pb['state'], seed=1, call=20 + i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 12
    tiles = 5 * ((N + 15) // 16)
    fl = 2 * (62 * units + 3 * units * units + 2 * units * 60) * 5 * N * 30 * 5
    print('units %d N %5d tiles %4d (%.2f per CU): plan %.3f ms, %.1f TFLOP/s, %.1f us per tile-step per CU' %
          (units, N, tiles, tiles / 256, dt * 1e3, fl / dt / 1e12, dt * 1e6 / 150 / (tiles / 256)), flush=True)
    pl.close()
