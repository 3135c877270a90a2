#!/usr/bin/env python3
"""Diagnostic: when the pinned tiles and the floating tiles' LAST segments of one rollout launch end (in-kernel s_memtime stamps).
Needs a library built with -DCEM_STAMPS (CEM_MPC_LIB=...); never used for timing claims."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ethz_safe_learning_amd import CemPlanner, PlannerConfig, synthetic
obs, act, H, I, K, N = 60, 2, 30, 5, 5, int(os.environ.get('CEM_N', '2000'))
pb = synthetic.problem(obs, act, K)
cfg = PlannerConfig(obs_dim=obs, act_dim=act, ensemble_size=K, particles=K, n_samples=N, horizon=H, n_elite=N // 10, iterations=I,
                    scorer=pb['scorer'], act_low=pb['low'], act_high=pb['high'], noise_stddev=1e-3, stddev_threshold=-1.0, use_graph=False)
pl = CemPlanner(cfg); pl.set_weights(pb['weights']); pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
for i in range(3):
    pl.plan(pb['state'], seed=1, call=i)
rc, tiles = pl.tiles(); nt = len(tiles)
nseg, seglen = pl.segments()
st = pl._view(pl.layout.stamps, nt * 4 * 8, torch.int64).view(nt, 4, 8).cpu().numpy().astype(np.float64)
npin = (nt // 256) * 256 if nseg > 1 else nt
start = st[:, 1, 7]; dur = st[:, 1, :5].sum(axis=1) + st[:, 1, 5:7].sum(axis=1) * 0     # wave 1: not the select kernel's slot
end = start + st[:, 1, :5].sum(axis=1)
xcc = ((st[:, 1, 6].astype(np.int64) & 15) << 8) | ((st[:, 1, 5].astype(np.int64) >> 8) & 255)   # (XCD, SE / SH / CU of HW_ID)
xcc = np.unique(xcc, return_inverse=True)[1]
                   # the stamps of different CUs do not share a zero: take every CU's own first pinned start as its zero
for x in np.unique(xcc):
    m = xcc == x
    z = start[m & (np.arange(nt) < npin)].min() if (m & (np.arange(nt) < npin)).any() else start[m].min()
    start[m] -= z; end[m] -= z
t0 = 0.0
pct = lambda a: np.percentile(a, [0, 10, 50, 90, 100]).round()
print('tiles %d (pinned %d, floating %d in %d segments of %d steps)' % (nt, npin, nt - npin, nseg, seglen))
print('pinned: start %s' % pct(start[:npin] - t0)); print('pinned: end   %s' % pct(end[:npin] - t0))
if nt > npin:
    steps_last = H - (nseg - 1) * seglen
    print('floating, last segment (%d steps): start %s' % (steps_last, pct(start[npin:] - t0)))
    print('floating, last segment: end   %s' % pct(end[npin:] - t0))
    print('floating, last segment: ticks per step %s' % pct((end[npin:] - start[npin:]) / steps_last))
print('pinned: ticks per step %s' % pct((end[:npin] - start[:npin]) / H))
print('launch span %.0f ticks; last pinned end %.0f, last floating end %.0f' % (end.max() - t0, end[:npin].max(), end[npin:].max() if nt > npin else 0))
