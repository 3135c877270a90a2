"""The fast CPU restatement (oracle/cem_oracle_fast.py, torch-CPU fp32 with one batched matmul over members per layer:
the CPU baseline of SURVEY.md 8d) against the numpy oracle and the golden fixtures.  Two independently written
restatements agreeing pins neither to the reference (PARITY UNPINNED: the reference holds no fixtures, TensorFlow is not
importable); it does catch a slip in either."""
import numpy as np
import pytest

from oracle import cem_oracle as o
from oracle import cem_oracle_fast as of
from tests import helpers as hp
from tests.test_golden import load


@pytest.mark.parametrize('variant', ['cem', 'safe'])
def test_fast_oracle_reproduces_golden(variant):
    z, pb, cfg = load(variant)
    tr = []
    a, s, it = of.plan(pb['state'], of.stack_weights(pb['weights']), pb['inputs_min'], pb['inputs_max'], pb['low'], pb['high'],
                       z['eps_act'], z['eps_model'], z['eps_out'], cfg, pb['scorer'], trace=tr)
    assert it == int(z['f32_iters'])
    np.testing.assert_allclose(a, z['f32_action'], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(a, z['f64_action'], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize('variant,P,E,case', [('cem', 5, 5, None), ('safe', 5, 5, None), ('cem', 5, 15, None), ('safe', 6, 3, None),
                                              ('safe', 5, 5, 'four_kinds_sum'), ('cem', 5, 5, 'goal_dist'),
                                              ('cem', 5, 5, 'active_reward_clip'), ('safe', 5, 5, 'no_reward_clip')])
def test_fast_oracle_matches_numpy_oracle(variant, P, E, case):
    pb = hp.scorer_problem(case, 60, E=E) if case else hp.make_problem(E=E, seed=31)
    N, H, I = 60 if E != 15 else 150, 7, 3
    ocfg, _ = hp.configs(pb, N=N, H=H, P=P, E=E, k=6, I=I, variant=variant, post=0.5, noise=0.01, smoothing=0.1)
    ea, em, eo = hp.noise(I, N, H, 2, P, 60, seed=4)
    t1, t2 = [], []
    a1, s1, i1 = o.do_generate_action(pb['state'], pb['weights'], pb['inputs_min'], pb['inputs_max'], pb['low'], pb['high'],
                                      ea, em, eo, ocfg, pb['scorer'], trace=t1)
    a2, s2, i2 = of.plan(pb['state'], of.stack_weights(pb['weights']), pb['inputs_min'], pb['inputs_max'], pb['low'], pb['high'],
                         ea, em, eo, ocfg, pb['scorer'], trace=t2)
    assert i1 == i2
    sc64 = None
    for it in range(i1):
        np.testing.assert_array_equal(t1[it]['actions'], t2[it]['actions']) if it == 0 else None
        bad = np.abs(t1[it]['scores'] - t2[it]['scores']) > 2e-5
        assert bad.mean() <= 0.05, (it, bad.sum())                      # a row on a `<=` threshold may flip in one of them
        if set(t1[it]['elite'].tolist()) != set(t2[it]['elite'].tolist()):
            pytest.skip('a near-tie on the k-th score flipped between the two fp32 summation orders')
        np.testing.assert_allclose(t1[it]['mu'], t2[it]['mu'], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(t1[it]['sigma'], t2[it]['sigma'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(a1, a2, rtol=1e-5, atol=1e-7)
    assert abs(s1 - s2) <= 2e-5
