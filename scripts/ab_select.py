#!/usr/bin/env python3
"""Same-box A/B of the select launch inside a whole B2 plan (HIP events on the planner's stream, eager launches): usage ab_select.py a.so b.so ..."""
import json, os, subprocess, sys
CODE = r'''
import json, sys, os
sys.path.insert(0, os.getcwd())
from ethz_safe_learning_amd import CemPlanner, PlannerConfig, synthetic
pb = synthetic.problem(60, 2, 5)
out = {}
for variant in ('cem', 'safe'):
    cfg = PlannerConfig(obs_dim=60, act_dim=2, ensemble_size=5, particles=5, n_samples=2000, horizon=30, n_elite=200, iterations=5, scorer=pb['scorer'],
                        act_low=pb['low'], act_high=pb['high'], noise_stddev=1e-3, variant=variant, posterior_mean_threashold=0.3)
    pl = CemPlanner(cfg); pl.set_weights(pb['weights']); pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
    for i in range(10): pl.plan(pb['state'], seed=1, call=i)
    pl.set_timing(True)
    sel = roll = n = 0
    for i in range(20):
        pl.plan(pb['state'], seed=2, call=i); tm = pl.last_timing(); sel += tm['select_ms']; roll += tm['rollout_ms']; n += tm['rollout_launches']
    out[variant] = dict(select_us=1e3 * sel / n, rollout_us=1e3 * roll / n)
    pl.close()
print(json.dumps(out))
'''
libs = [a for a in sys.argv[1:] if a.endswith('.so')]
for r in range(2):
    for l in libs:
        o = subprocess.run([sys.executable, '-c', CODE], env=dict(os.environ, CEM_MPC_LIB=os.path.abspath(l)), capture_output=True, text=True)
        print(os.path.basename(l), o.stdout.strip().splitlines()[-1] if o.stdout.strip() else o.stderr[-500:], flush=True)
