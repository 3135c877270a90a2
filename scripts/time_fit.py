import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_simba_api import make_agent_parts
np.random.seed(0)
env, model, pol = make_agent_parts('safe_cem_mpc', seed=1)      # E=15, units 128, 4 layers, batch 64, 5000 steps (config/models.yaml)
rng = np.random.default_rng(0)
n = 30000                                                         # train_batch_size (config/agents.yaml:7)
obs = rng.normal(0, 0.3, (n, 60)).astype(np.float32)
act = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
A = rng.normal(0, 0.02, (62, 60)).astype(np.float32)
nxt = obs + np.concatenate([obs, act], 1) @ A + 0.002 * rng.normal(0, 1, (n, 60)).astype(np.float32)
x = np.concatenate([obs, act], 1)
model.fit(x[:2000], nxt[:2000]) if False else None
t0 = time.perf_counter(); losses = model.fit(x, nxt); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('MlpEnsemble.fit: E=15, 5000 steps x batch 64, 30000 transitions: %.2f s (%.1f us/step); loss %.3f -> %.3f' % (dt, dt / 5000 * 1e6, losses[:50].mean(), losses[-50:].mean()))
t0 = time.perf_counter(); a = pol.generate_action(obs[0]); dt = time.perf_counter() - t0
print('first generate_action after fit (handle build + weight staging): %.3f s' % dt)
t0 = time.perf_counter()
for i in range(20): a = pol.generate_action(obs[i])
print('safe_cem_mpc shipped config (N=500,P=45,E=15,H=8,I<=9): %.2f ms / action' % ((time.perf_counter() - t0) / 20 * 1e3))
