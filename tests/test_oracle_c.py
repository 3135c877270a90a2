"""oracle/cem_oracle_c.c (the C + OpenMP restatement used as bench.py's CPU baseline) pinned to oracle/cem_oracle.py: whole plans on
identical explicit noise tensors — per-iteration candidate scores, iterations run, best score, action — over both objectives, smoothing,
early stop, goal-distance mode, several cost kinds, the non-indicator cost, propagation without sampling and members that split particles."""
import numpy as np
import pytest

from oracle import cem_oracle as o
from oracle import cem_oracle_c as oc

CASES = [
    dict(name='cem', variant='cem'),
    dict(name='safe', variant='safe', post=0.3),
    dict(name='safe_smoothing_stop', variant='safe', post=0.5, smoothing=0.3, thr=0.45, I=6),
    dict(name='cem_no_sampling_no_scale', variant='cem', sampling=False, scale=False),
    dict(name='safe_three_kinds_sum', variant='safe', post=0.3, kinds=3, indicator=False),
    dict(name='cem_goal_dist', variant='cem', goal_lidar=False),
    dict(name='cem_members_split_particles', variant='cem', E=3, P=2, N=30),
    dict(name='safe_shipped_shape', variant='safe', post=0.15, E=15, P=45, N=40, H=4, k=4),
]


@pytest.mark.parametrize('case', CASES, ids=[c['name'] for c in CASES])
def test_c_restatement_matches_the_numpy_oracle(case):
    E, P, N = case.get('E', 5), case.get('P', 5), case.get('N', 48)
    H, k, I = case.get('H', 6), case.get('k', 6), case.get('I', 3)
    pb = o.synthetic_problem(obs_dim=60, act_dim=2, ensemble_size=E, units=64, n_layers=3, seed=5)
    sp = pb['scorer']
    if case.get('kinds'):
        sp.cost_kinds = [(22, 38, 0.2), (41, 57, 0.35), (3, 19, 0.1)][:case['kinds']]
    if 'indicator' in case:
        sp.constrain_indicator = case['indicator']
    if case.get('goal_lidar') is False:
        sp.observe_goal_lidar = False
        sp.goal_slice = (0, 1)
    cfg = o.PlanConfig(horizon=H, iterations=I, n_samples=N, n_elite=k, particles=P, ensemble_size=E, smoothing=case.get('smoothing', 0.0),
                       stddev_threshold=case.get('thr', -1.0), noise_stddev=0.05, variant=case['variant'],
                       posterior_mean_threashold=case.get('post', 0.15), scale_features=case.get('scale', True),
                       sampling_propagation=case.get('sampling', True))
    rng = np.random.default_rng(9)
    ea = rng.standard_normal((I, N, H, 2)).astype(np.float32)
    em = rng.standard_normal((I, H, P * N, 60)).astype(np.float32)
    eo = rng.standard_normal(2).astype(np.float32)
    trace = []
    ra, rs, rit = o.do_generate_action(pb['state'], pb['weights'], pb['inputs_min'], pb['inputs_max'], pb['low'], pb['high'], ea, em, eo, cfg, sp, trace=trace)
    a, s, it, scores = oc.do_generate_action(pb['state'], pb['weights'], pb['inputs_min'], pb['inputs_max'], pb['low'], pb['high'], ea, em, eo, cfg, sp,
                                             return_scores=True)
    assert it == rit
    for i in range(it):
        ref = trace[i]['scores']
        # a candidate on a `<=` threshold may flip with the summation order; everything else agrees to fp32 rounding
        close = np.abs(scores[i] - ref) <= 2e-5 * np.maximum(1.0, np.abs(ref))
        assert close.mean() >= 0.97, (i, np.abs(scores[i] - ref).max())
    assert abs(s - rs) <= 2e-5 * max(1.0, abs(rs))
    np.testing.assert_allclose(a, ra, rtol=1e-4, atol=1e-5)


def test_c_restatement_draws_its_own_noise_and_uses_threads():
    pb = o.synthetic_problem(obs_dim=60, act_dim=2, ensemble_size=5, units=128, n_layers=4, seed=1)
    cfg = o.PlanConfig(horizon=10, iterations=2, n_samples=200, n_elite=20, particles=5, ensemble_size=5)
    a1, s1, it1 = oc.do_generate_action(pb['state'], pb['weights'], pb['inputs_min'], pb['inputs_max'], pb['low'], pb['high'], None, None, None, cfg,
                                        pb['scorer'], seed=3)
    a2, s2, it2 = oc.do_generate_action(pb['state'], pb['weights'], pb['inputs_min'], pb['inputs_max'], pb['low'], pb['high'], None, None, None, cfg,
                                        pb['scorer'], seed=3)
    assert it1 == it2 == 2 and np.isfinite(s1) and np.all(np.isfinite(a1))
    np.testing.assert_array_equal(a1, a2)                  # block-keyed generator: independent of the thread schedule
    assert oc.max_threads() >= 1
