import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ethz_safe_learning_amd import CemPlanner, PlannerConfig, synthetic
obs, act, K, H, I, N = 60, 2, 5, 30, 5, 2000
pb = synthetic.problem(obs, act, K)
cfg = PlannerConfig(obs_dim=obs, act_dim=act, ensemble_size=K, particles=K, n_samples=N, horizon=H, n_elite=N // 10, iterations=I,
                    scorer=pb['scorer'], act_low=pb['low'], act_high=pb['high'], noise_stddev=1e-3, chunks_per_tile=3)
pl = CemPlanner(cfg); pl.set_weights(pb['weights']); pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
for i in range(3):
    pl.plan(pb['state'], seed=1, call=i)
rc, tiles = pl.tiles(); nt = len(tiles)
st = pl._view(pl.layout.stamps, nt * 4 * 8, torch.int64).view(nt, 4, 8).cpu().numpy().astype(np.float64) / H
print('hidden stages (x3 per step), cycles per group F=0..7 (24 MFMA each = 768 ideal), waves 0..3, mean over tiles')
for w in range(4):
    print('wave %d:' % w, ' '.join('%6.0f' % st[:, w, F].mean() for F in range(8)), '  sum %.0f' % st[:, w, :].sum(axis=1).mean())
