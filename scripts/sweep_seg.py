#!/usr/bin/env python3
"""Diagnostic: rollout launch time at B2 (or CEM_N / CEM_H) over tile size x horizon segments."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ethz_safe_learning_amd import CemPlanner, PlannerConfig, synthetic
N = int(os.environ.get('CEM_N', '2000')); H = int(os.environ.get('CEM_H', '30')); K = int(os.environ.get('CEM_K', '5'))
P = int(os.environ.get('CEM_P', str(K)))                  # particles (default = members); the shipped safe_cem_mpc: CEM_N=500 CEM_H=8 CEM_K=15 CEM_P=45
pb = synthetic.problem(60, 2, K)
for rc in [int(x) for x in os.environ.get('CEM_RCS', '1,2,3,4').split(',')]:
    for seg in [int(x) for x in os.environ.get('CEM_SEGS', '1,3,6,10').split(',')]:
        cfg = PlannerConfig(obs_dim=60, act_dim=2, ensemble_size=K, particles=P, n_samples=N, horizon=H, n_elite=N // 10, iterations=5,
                            scorer=pb['scorer'], act_low=pb['low'], act_high=pb['high'], noise_stddev=1e-3, use_graph=False,
                            chunks_per_tile=rc, rollout_segments=seg)
        pl = CemPlanner(cfg); pl.set_weights(pb['weights']); pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
        for i in range(3):
            pl.plan(pb['state'], seed=1, call=i)
        pl.set_timing(True)
        ms, ln = 0.0, 0
        for i in range(6):
            pl.plan(pb['state'], seed=2, call=i); tm = pl.last_timing(); ms += tm['rollout_ms']; ln += tm['rollout_launches']
        print(json.dumps(dict(N=N, H=H, E=K, P=P, rc=rc, tiles=len(pl.tiles()[1]), segments=pl.segments()[0], rollout_ms=round(ms / ln, 4),
                              frac=round(synthetic.flops_per_row_step(60, 2) * P * N * H / (ms / ln * 1e-3) / 157.3e12, 3))), flush=True)
        pl.close(); del pl
