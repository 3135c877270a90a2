"""Fast CPU form of the oracle: the same path as oracle/cem_oracle.py restated on torch-CPU fp32 tensors, with the
ensemble evaluated as ONE batched matmul over members per layer (``torch.baddbmm``) instead of a Python loop over
members.  This is the CPU baseline SURVEY.md section 8d defines ("torch-CPU fp32 batched bmm over members").

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE — the same rule as cem_oracle.py: only ``tests/`` and the ``cpu_baseline``
leg of ``bench.py`` import it.  PARITY UNPINNED, as for cem_oracle.py: the reference ships no fixtures for this path and
TensorFlow is not importable here; this file is an independent second restatement (softplus, top-k, moments, the two
done-masking orders are written again from the reference, not shared with the numpy file) and tests/test_oracle_fast.py
checks the two against each other and against the golden fixtures.

Reference lines restated (relative to /root/reference):
  simba/policies/cem_mpc.py:35-68            do_generate_action
  simba/policies/mpc_policy.py:26-57         compute_objective, sampling_params
  simba/policies/safe_cem_mpc.py:76-120      compute_objective, bayesian_safety_beta_inference
  simba/models/transition_model.py:64-87     unfold_sequences, scale
  simba/models/mlp_ensemble.py:59-61,122-132,189-193
  simba/environment_utils/safety_gym.py:110-119,140-192
"""
from __future__ import annotations

import math

import torch


def stack_weights(weights):
    """list of per-member dicts (Keras layout) -> dict of [E, in, out] / [E, 1, out] tensors."""
    f = lambda a: torch.as_tensor(a, dtype=torch.float32)
    L = len(weights[0]['W'])
    return dict(W=[torch.stack([f(w['W'][l]) for w in weights]) for l in range(L)],
                b=[torch.stack([f(w['b'][l]) for w in weights]).unsqueeze(1) for l in range(L)],
                W_mu=torch.stack([f(w['W_mu']) for w in weights]), b_mu=torch.stack([f(w['b_mu']) for w in weights]).unsqueeze(1),
                W_var=torch.stack([f(w['W_var']) for w in weights]), b_var=torch.stack([f(w['b_var']) for w in weights]).unsqueeze(1))


def softplus_tf(x):
    """tf.math.softplus (Eigen): x above -thr, exp(x) below thr, log1p(exp(x)) between; thr = log(eps_f32) + 2."""
    thr = math.log(torch.finfo(torch.float32).eps) + 2.0
    ex = torch.exp(x)
    return torch.where(x > -thr, x, torch.where(x < thr, ex, torch.log1p(ex)))


def ensemble_forward(x, sw):
    """MlpEnsemble.forward (mlp_ensemble.py:122-132): tf.split(x, E) -> member m gets rows [m*B/E, (m+1)*B/E)."""
    E = sw['W_mu'].shape[0]
    B = x.shape[0]
    if B % E:
        raise ValueError('tf.split requires B % E == 0')
    h = x.view(E, B // E, x.shape[1])
    for W, b in zip(sw['W'], sw['b']):
        h = torch.relu(torch.baddbmm(b, h, W))                               # Dense + ReLU (mlp_ensemble.py:18-22)
    mu = torch.baddbmm(sw['b_mu'], h, sw['W_mu'])
    var = softplus_tf(torch.baddbmm(sw['b_var'], h, sw['W_var'])) + 1e-4     # GaussianHead (mlp_ensemble.py:33-34)
    return mu.reshape(B, -1), var.reshape(B, -1)


def closest_distance(lidar, D):
    return torch.clamp(D - D * (1.0 - lidar), 0.0, D).amin(dim=1)            # safety_gym.py:188-192


def goal_distance(obs, sp):
    lo, hi = sp.goal_slice
    if sp.observe_goal_lidar:
        return closest_distance(obs[:, lo:hi], float(sp.lidar_max_dist))
    return torch.relu(obs[:, lo])                                            # safety_gym.py:172-174


def cost(obs, sp):
    c = torch.zeros(obs.shape[0])
    for lo, hi, size in sp.cost_kinds:                                        # safety_gym.py:148-163
        c = c + (closest_distance(obs[:, lo:hi], float(sp.lidar_max_dist)) <= size).float()
    return (c > 0).float() if sp.constrain_indicator else c


def plan(state, sw, inputs_min, inputs_max, low, high, eps_act, eps_model, eps_out, cfg, sp, trace=None):
    """CemMpc.do_generate_action (cem_mpc.py:35-68) with compute_objective of cfg.variant.  Noise tensors as in
    cem_oracle.do_generate_action; eps_model may be None (fresh torch.randn per step: timing mode)."""
    f32 = torch.float32
    state = torch.as_tensor(state, dtype=f32)
    low, high = torch.as_tensor(low, dtype=f32), torch.as_tensor(high, dtype=f32)
    if bool(torch.isfinite(low).all() and torch.isfinite(high).all()):       # mpc_policy.py:45-57
        lb, ub, mu0, sg0 = low, high, (high + low) / 2.0, (high - low) / 2.0
    else:
        A0 = low.shape[0]
        lb, ub, mu0, sg0 = torch.full((A0,), -100.0), torch.full((A0,), 100.0), torch.zeros(A0), torch.full((A0,), 100.0)
    H, N, P, k, A, O = cfg.horizon, cfg.n_samples, cfg.particles, cfg.n_elite, low.shape[0], state.shape[0]
    B = P * N
    imin, imax = torch.as_tensor(inputs_min, dtype=f32), torch.as_tensor(inputs_max, dtype=f32)
    delta = imax - imin
    delta = torch.where(delta < 1e-5, torch.tensor(1.01), delta)            # transition_model.py:84-85
    mu, sigma = mu0.expand(H, A).clone(), sg0.expand(H, A).clone()
    best, best_score = torch.zeros(A), torch.tensor(-math.inf)
    thr_goal = float(torch.tensor(sp.goal_size * 0.8, dtype=f32))            # python float * 0.8 -> fp32 tensor (safety_gym.py:116)
    safe_variant = cfg.variant == 'safe'
    if safe_variant:                                                         # safe_cem_mpc.py:113-115, fp32 tensor arithmetic
        m_, s_ = torch.tensor(0.5), torch.tensor(0.27)
        alpha = (((1.0 - m_) / s_ ** 2) - 1.0 / m_) * m_ ** 2
        beta = alpha * (1.0 / m_ - 1.0)
    iters = 0
    for it in range(cfg.iterations):
        a = torch.clamp(torch.as_tensor(eps_act[it], dtype=f32) * sigma + mu, lb, ub)         # cem_mpc.py:44-48
        a_b = a.repeat(P, 1, 1)                                                               # tf.tile (cem_mpc.py:49-51)
        s = state.expand(B, O).clone()
        cum, done = torch.zeros(B), torch.zeros(B, dtype=torch.bool)
        safe = torch.ones(N, dtype=torch.bool)
        d = goal_distance(s, sp)
        for t in range(H):                                                   # transition_model.py:69-75 fused with the objective loop
            x = torch.cat([s, a_b[:, t]], dim=1)
            if cfg.scale_features:
                x = (x - imin) / delta
            m, var = ensemble_forward(x, sw)
            if cfg.sampling_propagation:
                e = torch.as_tensor(eps_model[it][t], dtype=f32) if eps_model is not None else torch.randn(B, O)
                s_next = s + (m + torch.sqrt(var) * e)                       # Normal.sample = loc + scale * eps
            else:
                s_next = s + m
            dn = goal_distance(s_next, sp)
            ga = d <= thr_goal
            r = (d - dn) * sp.reward_distance + ga.float() * sp.reward_goal
            if sp.reward_clip:
                r = torch.clamp(r, -sp.reward_clip, sp.reward_clip)
            if safe_variant:                                                 # safe_cem_mpc.py:86-93: done first
                done = done | ga
                nd = 1.0 - done.float()
                c = cost(s, sp) * nd
                counts = c.view(P, N).sum(dim=0)
                post = (alpha + counts) / (alpha + beta + float(P))
                safe = safe & (post <= cfg.posterior_mean_threashold)
                cum = cum + r * nd
            else:                                                            # mpc_policy.py:34-37: reward first
                cum = cum + r * (1.0 - done.float())
                done = done | ga
            s, d = s_next, dn
        scores = cum.view(P, N).sum(dim=0) / float(P)
        if safe_variant:
            scores = scores - (~safe).float() * 100.0
        # tf.nn.top_k(sorted=False): the k largest, ties -> lower index; kept in ascending index order
        order = torch.argsort(scores, descending=True, stable=True)[:k]
        elite = torch.sort(order).values
        j = elite[torch.argmax(scores[elite])]
        if scores[j] > best_score:                                           # strict (cem_mpc.py:58)
            best, best_score = a[j, 0].clone(), scores[j].clone()
        el = a[elite]
        mean = el.sum(dim=0) / float(k)                                      # tf.nn.moments: population variance
        var_e = ((el - mean) ** 2).sum(dim=0) / float(k)
        # cem_mpc.py:64-65: both factors are Python floats rounded once to fp32 (torch converts a Python scalar the same way)
        s32 = float(torch.tensor(float(cfg.smoothing), dtype=torch.float32))
        oms32 = float(torch.tensor(1.0 - float(cfg.smoothing), dtype=torch.float32))
        mu = s32 * mu + oms32 * mean
        sigma = s32 * sigma + oms32 * torch.sqrt(var_e)
        iters += 1
        if trace is not None:
            trace.append(dict(actions=a.numpy().copy(), scores=scores.numpy().copy(), elite=elite.numpy().copy(),
                              mu=mu.numpy().copy(), sigma=sigma.numpy().copy()))
        if float(sigma.mean()) <= cfg.stddev_threshold:                      # cem_mpc.py:66-67
            break
    action = best + torch.as_tensor(eps_out, dtype=f32) * cfg.noise_stddev   # cem_mpc.py:68
    return action.numpy(), float(best_score), iters
