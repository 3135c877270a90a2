#!/usr/bin/env python3
"""Diagnostic: per-step cycle attribution of cem_rollout_kernel from in-kernel s_memtime stamps.
Needs a library built with -DCEM_STAMPS (CEM_MPC_LIB=...); never used for timing claims."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ethz_safe_learning_amd import CemPlanner, PlannerConfig, synthetic
obs, act, H, I = int(os.environ.get('CEM_OBS', '60')), int(os.environ.get('CEM_ACT', '2')), int(os.environ.get('CEM_H', '30')), 5
K, N = int(os.environ.get('CEM_K', '5')), int(os.environ.get('CEM_N', '2000'))
pb = synthetic.problem(obs, act, K)
chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cfg = PlannerConfig(obs_dim=obs, act_dim=act, ensemble_size=K, particles=K, n_samples=N, horizon=H, n_elite=N // 10, iterations=I,
                    scorer=pb['scorer'], act_low=pb['low'], act_high=pb['high'], noise_stddev=1e-3, chunks_per_tile=chunks, rollout_segments=1,
                    precision=os.environ.get('CEM_PRECISION', 'fp32'))
pl = CemPlanner(cfg); pl.set_weights(pb['weights']); pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
for i in range(3):
    pl.plan(pb['state'], seed=1, call=i)
rc, tiles = pl.tiles()
nt = len(tiles)
st = pl._view(pl.layout.stamps, nt * 4 * 8, torch.int64).view(nt, 4, 8).cpu().numpy().astype(np.float64) / H
names = ['layer0 stage', 'hidden stages', 'heads pre (loads+philox)', 'heads MFMA', 'epilogue', 'barrier wait', 'bookkeep+readX', '-']
if os.environ.get('CEM_PRECISION', 'fp32') != 'fp32':
    names[5], names[6] = 'relu + split + publish', 'bookkeeping'
st = np.delete(st, 2, axis=0) if nt > 3 else st      # tile 2 / wave 0 shares its slot with the select kernel's stamps
nt = st.shape[0]
print('chunks/tile %d, tiles %d; cycles per step (s_memtime ticks), mean over tiles; waves 0..3' % (rc, nt))
for i, n in enumerate(names[:7]):
    print('%-26s' % n, ' '.join('%8.0f' % st[:, w, i].mean() for w in range(4)))
print('%-26s' % 'sum', ' '.join('%8.0f' % st[:, w, :7].sum(axis=1).mean() for w in range(4)))
start = st[:, 0, 7] * H
dur = st[:, 0, :7].sum(axis=1) * H
t0 = start.min()
print('tile start offsets (cycles): pct 0/50/90/100 = %s' % np.percentile(start - t0, [0, 50, 90, 100]).round())
print('tile durations (cycles):     pct 0/10/50/90/100 = %s' % np.percentile(dur, [0, 10, 50, 90, 100]).round())
print('kernel span (first start -> last end) = %.0f cycles = %.3f ms @2.4GHz' % ((start + dur).max() - t0, ((start + dur).max() - t0) / 2.4e6))
late = (start - t0) > 0.2 * dur.mean()
print('tiles starting late (second round): %d of %d' % (late.sum(), nt))
