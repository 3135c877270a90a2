#!/usr/bin/env python3
"""Diagnostic: where cem_select_kernel's time goes (s_memtime stamps of a -DCEM_STAMPS build, CEM_MPC_LIB=...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ethz_safe_learning_amd import CemPlanner, PlannerConfig, synthetic
pb = synthetic.problem(60, 2, 5)
names = ['prologue', 'stage scores', 'radix select', 'compaction', 'best-of-elite', 'moments', 'tail']
VARIANT = os.environ.get('CEM_VARIANT', 'cem')          # 'safe': SafeCemMpc scores (most candidates near -100), k = 4 % as the reference ships it
H = int(os.environ.get('CEM_H', '30'))
for N in [int(a) for a in sys.argv[1:]] or [2000, 16000]:
    k = int(os.environ.get('CEM_K', '0')) or max(1, round(N / (25 if VARIANT == 'safe' else 10)))
    cfg = PlannerConfig(obs_dim=60, act_dim=2, ensemble_size=5, particles=5, n_samples=N, horizon=H, n_elite=k, iterations=5, world_size=int(os.environ.get("CEM_W", "1")),
                        scorer=pb['scorer'], act_low=pb['low'], act_high=pb['high'], noise_stddev=1e-3, use_graph=False, variant=VARIANT,
                        posterior_mean_threashold=float(os.environ.get('CEM_POST', '0.3')))
    pl = CemPlanner(cfg); pl.set_weights(pb['weights']); pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
    acc = []
    for i in range(int(os.environ.get('CEM_PLANS', '8'))):
        pl.plan_begin(pb['state'], seed=1, call=i)          # stepwise: also valid for a shard handle (CEM_W > 1) without a communicator
        for it in range(5):
            pl.plan_rollout(it); pl.plan_select(it)
            if i >= 2:                                       # every select of the later plans: one launch's stamps are +-0.5 us
                pl.stream.synchronize()
                acc.append(pl._view(pl.layout.stamps + 64 * 8, 8, torch.int64).cpu().numpy().astype(np.float64))
        pl.plan_end()
    st = np.mean(np.stack([a - a[0] for a in acc]), axis=0)
    print('%s N %6d k %d H %d:' % (VARIANT, N, k, H), '  '.join('%s %.2f' % (n, (st[i + 1] - st[i]) / 2400.0) for i, n in enumerate(names[1:])),
          ' total %.1f us @2.4 GHz  (mean of %d selects)' % ((st[6] - st[0]) / 2400.0, len(acc)))
    pl.close()
