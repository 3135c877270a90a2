#!/usr/bin/env python3
"""One of bench.py's policy legs (POLICY_LEGS: the reference's default policy and shipped shapes) or BASELINE configs by name, as
plain graph-replayed plans — the program rocprofv3 wraps for profiles/r05_kernel_stats_<leg>.csv (scripts/collect_profiles_r05.sh).
usage: CEM_LEG=B2_safe python3 scripts/run_policy_leg.py [plans]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np   # noqa: E402
import torch         # noqa: E402

import bench         # noqa: E402
from ethz_safe_learning_amd import CemPlanner, PlannerConfig, synthetic   # noqa: E402

leg = os.environ.get('CEM_LEG', 'B2_safe')
plans = int(sys.argv[1]) if len(sys.argv) > 1 else 40
row = next(r for r in bench.POLICY_LEGS if r[0] == leg)
name, K, P, N, H, I, k, variant, thr, post, noise, _, _ = row
pb = synthetic.problem(60, 2, K)
cfg = PlannerConfig(obs_dim=60, act_dim=2, ensemble_size=K, particles=P, n_samples=N, horizon=H, n_elite=k, iterations=I,
                    scorer=pb['scorer'], act_low=pb['low'], act_high=pb['high'], stddev_threshold=thr, noise_stddev=noise, variant=variant,
                    posterior_mean_threashold=post, use_graph=not os.environ.get('CEM_LEG_NOGRAPH'), select_mode=int(os.environ.get('CEM_LEG_SELECT_MODE', '0')))
pl = CemPlanner(cfg)
pl.set_weights(pb['weights'])
pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
for i in range(15):
    pl.plan(pb['state'], seed=2029, call=i)
torch.cuda.synchronize()
t0 = time.perf_counter()
its = []
for i in range(plans):
    a, s, it = pl.plan(pb['state'], seed=2029, call=15 + i)
    its.append(it)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps(dict(leg=leg, plans=plans, ms_per_plan=1e3 * dt / plans, plans_per_s=plans / dt, iterations_run_mean=float(np.mean(its)),
                      launches_per_iteration=pl.launches_per_iteration(), tiles=int(len(pl.tiles()[1])), chunks_per_tile=pl.tiles()[0],
                      graph=pl.graph_status())), flush=True)
pl.close()
