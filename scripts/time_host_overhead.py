#!/usr/bin/env python3
"""Where a B2 plan's wall time goes on the host side: CemPlanner.plan() vs the bare C call (ctypes), per plan, graph replay."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ethz_safe_learning_amd import CemPlanner, PlannerConfig, synthetic, _capi
pb = synthetic.problem(60, 2, 5)
cfg = PlannerConfig(obs_dim=60, act_dim=2, ensemble_size=5, particles=5, n_samples=2000, horizon=30, n_elite=200, iterations=5, scorer=pb['scorer'],
                    act_low=pb['low'], act_high=pb['high'], noise_stddev=1e-3, use_graph=True)
pl = CemPlanner(cfg); pl.set_weights(pb['weights']); pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
for i in range(30): pl.plan(pb['state'], seed=1, call=i)
n = 300
t0 = time.perf_counter()
for i in range(n): pl.plan(pb['state'], seed=1, call=100 + i)
t_py = (time.perf_counter() - t0) / n
st = np.ascontiguousarray(pb['state'], np.float32); action = np.zeros(2, np.float32); score = C.c_float(); iters = C.c_int32()
args = (pl.h, st.ctypes.data_as(C.c_void_p), 1, 0, None, None, None, action.ctypes.data_as(C.c_void_p), C.byref(score), C.byref(iters))
f = pl.lib.cem_planner_plan
t0 = time.perf_counter()
for i in range(n): f(*args)
t_c = (time.perf_counter() - t0) / n
print('plan() %.1f us   bare C call %.1f us   python wrapper %.1f us' % (t_py * 1e6, t_c * 1e6, (t_py - t_c) * 1e6))
