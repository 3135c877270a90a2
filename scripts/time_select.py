#!/usr/bin/env python3
"""Diagnostic: cem_select_kernel time vs the (global) candidate count, e.g. N = 2000 x G of the weak-scaled bench."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ethz_safe_learning_amd import CemPlanner, PlannerConfig, synthetic
pb = synthetic.problem(60, 2, 5)
import os
MODE = int(os.environ.get('CEM_SELECT_MODE', '0'))
for N in (500, 2000, 4000, 8000, 16000, 40000, 65536):
    cfg = PlannerConfig(obs_dim=60, act_dim=2, ensemble_size=5, particles=5, n_samples=N, horizon=30, n_elite=N // 10, iterations=5,
                        scorer=pb['scorer'], act_low=pb['low'], act_high=pb['high'], noise_stddev=1e-3, use_graph=False, select_mode=MODE)
    pl = CemPlanner(cfg); pl.set_weights(pb['weights']); pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
    for i in range(2):
        pl.plan(pb['state'], seed=1, call=i)
    pl.set_timing(True)
    sel = 0.0; n = 0
    for i in range(4):
        pl.plan(pb['state'], seed=2, call=i); tm = pl.last_timing(); sel += tm['select_ms']; n += tm['rollout_launches']
    print('mode %d  N %6d  k %5d  select %.1f us/launch' % (MODE, N, N // 10, 1e3 * sel / n))
    pl.close()
