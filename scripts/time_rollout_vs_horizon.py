#!/usr/bin/env python3
"""Rollout launch time against the horizon at the B2 shape (obs 60, act 2, K = P = E = 5, N = 2000): T(H) = start-up + H x per-step.
What a launch pays before / after its step loop (dispatch, tile descriptors, first weight groups, instruction cache, the tail where
CUs run with fewer resident tiles) shows as the intercept.  Plain launches (no floating segments) and the automatic plan."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ethz_safe_learning_amd import CemPlanner, PlannerConfig, synthetic

pb = synthetic.problem(60, 2, 5)
N = int(os.environ.get('N', '2000'))
for seg in (1, 0):
    for H in (1, 2, 4, 8, 15, 30, 60):
        cfg = PlannerConfig(obs_dim=60, act_dim=2, ensemble_size=5, particles=5, n_samples=N, horizon=H, n_elite=N // 10, iterations=2,
                            scorer=pb['scorer'], act_low=pb['low'], act_high=pb['high'], noise_stddev=1e-3, use_graph=False, rollout_segments=seg,
                            chunks_per_tile=int(os.environ.get('RC', '1')))
        pl = CemPlanner(cfg); pl.set_weights(pb['weights']); pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
        for i in range(6):
            pl.plan(pb['state'], seed=1, call=i)
        pl.set_timing(True)
        ms, ln = 0.0, 0
        for i in range(6):
            pl.plan(pb['state'], seed=2, call=i); tm = pl.last_timing(); ms += tm['rollout_ms']; ln += tm['rollout_launches']
        print(json.dumps(dict(N=N, H=H, segments=pl.segments()[0], tiles=int(len(pl.tiles()[1])), rollout_us=1e3 * ms / ln)), flush=True)
        pl.close(); del pl
