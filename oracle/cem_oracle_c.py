"""ctypes binding of oracle/cem_oracle_c.c — the C + OpenMP restatement of one CEM-MPC plan (TEST INFRASTRUCTURE: only tests/ and
bench.py's cpu_baseline leg import this).  Pinned to oracle/cem_oracle.py by tests/test_oracle_c.py; "parity unpinned" like it."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import cem_oracle as o

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'libcem_oracle_c.so')


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('O', 'A', 'U', 'L', 'E', 'P', 'N', 'H', 'k', 'I')] + \
               [(n, C.c_float) for n in ('smoothing', 'one_minus_smoothing', 'stddev_threshold', 'noise_stddev')] + \
               [('variant', C.c_int32), ('posterior_mean_threashold', C.c_float), ('scale_features', C.c_int32), ('sampling_propagation', C.c_int32),
                ('observe_goal_lidar', C.c_int32), ('goal_lo', C.c_int32), ('goal_hi', C.c_int32)] + \
               [(n, C.c_float) for n in ('lidar_max_dist', 'goal_thresh', 'reward_distance', 'reward_goal', 'reward_clip')] + \
               [('constrain_indicator', C.c_int32), ('n_cost', C.c_int32), ('cost_lo', C.c_int32 * 4), ('cost_hi', C.c_int32 * 4), ('cost_size', C.c_float * 4)]


_lib = None


def build():
    r = subprocess.run(['make', '-C', HERE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError('gcc build of oracle/libcem_oracle_c.so failed:\n' + r.stdout)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.cem_c_plan.restype = C.c_int
        _lib.cem_c_max_threads.restype = C.c_int
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def flatten_weights(weights):
    """The natural per-member blob of include/cem_mpc.h: W_0, b_0, ..., W_mu, b_mu, W_var, b_var."""
    parts = []
    for w in weights:
        for W, b in zip(w['W'], w['b']):
            parts += [np.asarray(W, np.float32).ravel(), np.asarray(b, np.float32).ravel()]
        parts += [np.asarray(w['W_mu'], np.float32).ravel(), np.asarray(w['b_mu'], np.float32).ravel(),
                  np.asarray(w['W_var'], np.float32).ravel(), np.asarray(w['b_var'], np.float32).ravel()]
    return np.ascontiguousarray(np.concatenate(parts))


def do_generate_action(state, weights, inputs_min, inputs_max, low, high, eps_act, eps_model, eps_out, cfg: o.PlanConfig, sp: o.ScorerParams,
                       seed=0, return_scores=False):
    """Same signature and result as cem_oracle.do_generate_action; eps_* = None draws the noise inside (timing mode)."""
    lib = load()
    w0 = weights[0]
    if str(w0.get('activation', 'relu')).split('.')[-1].lower() != 'relu':
        raise NotImplementedError('the C restatement is relu only')
    O, A = int(np.asarray(state).shape[0]), int(np.asarray(low).shape[0])
    c = Config()
    c.O, c.A, c.U, c.L, c.E = O, A, int(w0['W'][0].shape[1]), len(w0['W']), cfg.ensemble_size
    c.P, c.N, c.H, c.k, c.I = cfg.particles, cfg.n_samples, cfg.horizon, cfg.n_elite, cfg.iterations
    c.smoothing, c.one_minus_smoothing = float(np.float32(cfg.smoothing)), float(np.float32(1.0 - float(cfg.smoothing)))
    c.stddev_threshold, c.noise_stddev = float(cfg.stddev_threshold), float(cfg.noise_stddev)
    c.variant = 1 if cfg.variant == 'safe' else 0
    c.posterior_mean_threashold = float(cfg.posterior_mean_threashold)
    c.scale_features, c.sampling_propagation = int(cfg.scale_features), int(cfg.sampling_propagation)
    c.observe_goal_lidar, (c.goal_lo, c.goal_hi) = int(sp.observe_goal_lidar), sp.goal_slice
    c.lidar_max_dist, c.goal_thresh = float(sp.lidar_max_dist), float(np.float32(sp.goal_size * 0.8))
    c.reward_distance, c.reward_goal, c.reward_clip = float(sp.reward_distance), float(sp.reward_goal), float(sp.reward_clip or 0.0)
    c.constrain_indicator, c.n_cost = int(sp.constrain_indicator), len(sp.cost_kinds)
    for i, (lo, hi, size) in enumerate(sp.cost_kinds):
        c.cost_lo[i], c.cost_hi[i], c.cost_size[i] = lo, hi, float(size)
    lb, ub, mu0, sg0 = o.sampling_params(low, high, np.float32)
    blob = flatten_weights(weights)
    f32 = lambda a: None if a is None else np.ascontiguousarray(np.asarray(a, np.float32))
    st, imin, imax, lb, ub, mu0, sg0 = (f32(x) for x in (state, inputs_min, inputs_max, lb, ub, mu0, sg0))
    ea, em, eo = f32(eps_act), f32(eps_model), f32(eps_out)
    action = np.zeros(A, np.float32)
    score, iters = C.c_float(), C.c_int32()
    scores = np.zeros((cfg.iterations, cfg.n_samples), np.float32) if return_scores else None
    rc = lib.cem_c_plan(C.byref(c), _ptr(blob), _ptr(imin), _ptr(imax), _ptr(lb), _ptr(ub), _ptr(mu0), _ptr(sg0), _ptr(st), _ptr(ea), _ptr(em), _ptr(eo),
                        C.c_uint64(seed), _ptr(action), C.byref(score), C.byref(iters), _ptr(scores))
    assert rc == 0
    out = (action, np.float32(score.value), int(iters.value))
    return out + (scores,) if return_scores else out


def max_threads():
    return int(load().cem_c_max_threads())


def set_threads(n):
    load().cem_c_set_threads(C.c_int(int(n)))
