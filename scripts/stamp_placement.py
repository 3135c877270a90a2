#!/usr/bin/env python3
"""Diagnostic (-DCEM_STAMPS library): which CU ran each tile of one rollout launch, when it started and how long it took."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collections import Counter
from ethz_safe_learning_amd import CemPlanner, PlannerConfig, synthetic
K, N = int(os.environ.get('CEM_K', '5')), int(os.environ.get('CEM_N', '2000'))
H = 30
pb = synthetic.problem(60, 2, K)
cfg = PlannerConfig(obs_dim=60, act_dim=2, ensemble_size=K, particles=K, n_samples=N, horizon=H, n_elite=N // 10, iterations=1,
                    scorer=pb['scorer'], act_low=pb['low'], act_high=pb['high'], noise_stddev=1e-3, chunks_per_tile=int(sys.argv[1]) if len(sys.argv) > 1 else 1,
                    rollout_segments=1)
pl = CemPlanner(cfg); pl.set_weights(pb['weights']); pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
for i in range(3):
    pl.plan(pb['state'], seed=1, call=i)
nt = len(pl.tiles()[1])
st = pl._view(pl.layout.stamps, nt * 4 * 8, torch.int64).view(nt, 4, 8).cpu().numpy()
st = np.delete(st, 2, axis=0)
hw, xcc, start = st[:, 1, 5], st[:, 1, 6] & 0xf, st[:, 1, 7].astype(np.float64)
dur = st[:, 1, :5].sum(axis=1).astype(np.float64)
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = [(int(x), int(s), int(h), int(c)) for x, s, h, c in zip(xcc, se, sh, cu)]
cnt = Counter(key)
print('tiles %d, distinct CUs %d, tiles per CU histogram: %s' % (nt, len(cnt), sorted(Counter(cnt.values()).items())))
print('per XCD tiles:', sorted(Counter(int(x) for x in xcc).items()))
t0 = start.min()
per = {}
for k, s_, d in zip(key, start, dur):
    per.setdefault(k, []).append((s_ - t0, d))
for n in sorted(set(cnt.values())):
    ds = [d for k, v in per.items() if len(v) == n for (_, d) in v]
    ss = [s_ for k, v in per.items() if len(v) == n for (s_, _) in v]
    print('CUs with %d tiles: tile duration mean %.0f (min %.0f max %.0f) cycles, start offsets up to %.0f' % (n, np.mean(ds), np.min(ds), np.max(ds), np.max(ss)))
print('kernel span %.0f cycles' % ((start + dur).max() - t0))
