#!/usr/bin/env python3
"""Diagnostic: where cem_select_kernel's time goes (s_memtime stamps of a -DCEM_STAMPS build, CEM_MPC_LIB=...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ethz_safe_learning_amd import CemPlanner, PlannerConfig, synthetic
pb = synthetic.problem(60, 2, 5)
names = ['prologue', 'stage scores', 'radix select', 'compaction', 'best-of-elite', 'moments', 'tail']
for N in [int(a) for a in sys.argv[1:]] or [2000, 16000]:
    cfg = PlannerConfig(obs_dim=60, act_dim=2, ensemble_size=5, particles=5, n_samples=N, horizon=30, n_elite=max(1, round(N / 10)), iterations=5, world_size=int(os.environ.get("CEM_W", "1")),
                        scorer=pb['scorer'], act_low=pb['low'], act_high=pb['high'], noise_stddev=1e-3, use_graph=False)
    pl = CemPlanner(cfg); pl.set_weights(pb['weights']); pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
    for i in range(3):
        pl.plan_begin(pb['state'], seed=1, call=i)          # stepwise: also valid for a shard handle (CEM_W > 1) without a communicator
        for it in range(5):
            pl.plan_rollout(it); pl.plan_select(it)
        pl.plan_end()
    st = pl._view(pl.layout.stamps + 64 * 8, 8, torch.int64).cpu().numpy().astype(np.float64)
    print('N %6d:' % N, '  '.join('%s %.2f' % (n, (st[i + 1] - st[i]) / 2400.0) for i, n in enumerate(names[1:])),
          ' total %.1f us @2.4 GHz' % ((st[6] - st[0]) / 2400.0))
    pl.close()
