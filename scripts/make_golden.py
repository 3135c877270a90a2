#!/usr/bin/env python3
"""Generate tests/golden/*.npz: small input/output vectors for the CEM-MPC path.

The reference (TensorFlow) cannot run in this image and ships no fixtures, so these vectors come from
oracle/cem_oracle.py: float64 ("f64_*" arrays, the shadow) and float32 ("f32_*", the reference's arithmetic type).
They pin the oracle against regressions and give the GPU tests a fixed target that does not depend on the oracle code
at test time.  Regenerate with:  python scripts/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cem_oracle as o      # noqa: E402


def flat(weights):
    out = {}
    for m, w in enumerate(weights):
        for l, (W, b) in enumerate(zip(w['W'], w['b'])):
            out['m%d_W%d' % (m, l)] = W
            out['m%d_b%d' % (m, l)] = b
        for k in ('W_mu', 'b_mu', 'W_var', 'b_var'):
            out['m%d_%s' % (m, k)] = w[k]
    return out


def make(variant, path):
    O, A, E, U, L = 6, 2, 2, 128, 2
    N, H, P, k, I = 32, 4, 2, 4, 3
    pb = o.synthetic_problem(obs_dim=O, act_dim=A, ensemble_size=E, units=U, n_layers=L, seed=77)
    rng = np.random.default_rng(5)
    for w in pb['weights']:
        for b in w['b']:
            b[:] = rng.normal(0, 0.05, b.shape)
        w['b_mu'][:] = rng.normal(0, 0.01, O)
    cfg = o.PlanConfig(horizon=H, iterations=I, n_samples=N, n_elite=k, particles=P, ensemble_size=E, smoothing=0.09,
                       stddev_threshold=-1.0, noise_stddev=0.05, variant=variant, posterior_mean_threashold=0.45)
    ea = rng.standard_normal((I, N, H, A)).astype(np.float32)
    em = rng.standard_normal((I, H, P * N, O)).astype(np.float32)
    eo = rng.standard_normal((A,)).astype(np.float32)
    res = {}
    for name, dt in (('f64', np.float64), ('f32', np.float32)):
        tr = []
        a, s, it = o.do_generate_action(pb['state'], o.cast_weights(pb['weights'], dt), pb['inputs_min'], pb['inputs_max'],
                                        pb['low'], pb['high'], ea, em, eo, cfg, pb['scorer'], dtype=dt, trace=tr)
        res[name + '_action'] = a
        res[name + '_best_score'] = np.asarray(s)
        res[name + '_iters'] = np.asarray(it)
        res[name + '_scores'] = np.stack([t['scores'] for t in tr])
        res[name + '_elite'] = np.stack([t['elite'] for t in tr])
        res[name + '_mu'] = np.stack([t['mu'] for t in tr])
        res[name + '_sigma'] = np.stack([t['sigma'] for t in tr])
        res[name + '_actions'] = np.stack([t['actions'] for t in tr])
    sp = pb['scorer']
    np.savez_compressed(path, variant=variant, dims=np.array([O, A, E, U, L, N, H, P, k, I]), smoothing=0.09, noise_stddev=0.05,
                        posterior=0.45, state=pb['state'], inputs_min=pb['inputs_min'], inputs_max=pb['inputs_max'],
                        low=pb['low'], high=pb['high'], goal_slice=np.array(sp.goal_slice),
                        cost_kinds=np.array(sp.cost_kinds, np.float64), eps_act=ea, eps_model=em, eps_out=eo,
                        **flat(pb['weights']), **res)


if __name__ == '__main__':
    d = os.path.join(ROOT, 'tests', 'golden')
    os.makedirs(d, exist_ok=True)
    for v in ('cem', 'safe'):
        make(v, os.path.join(d, 'tiny_plan_%s.npz' % v))
        print('wrote', v)
