"""Shared builders for the parity tests: one synthetic problem -> (oracle config, planner config)."""
import os

import numpy as np

from oracle import cem_oracle as o

from ethz_safe_learning_amd import PlannerConfig, ScorerConfig


def make_problem(obs_dim=60, act_dim=2, E=5, n_layers=4, seed=1234, bias_noise=0.05, units=128, **kw):
    pb = o.synthetic_problem(obs_dim=obs_dim, act_dim=act_dim, ensemble_size=E, units=units, n_layers=n_layers, seed=seed, **kw)
    if bias_noise:
        rng = np.random.default_rng(seed + 999)
        for w in pb['weights']:
            for b in w['b']:
                b[:] = rng.normal(0, bias_noise, b.shape).astype(np.float32)
            w['b_mu'][:] = rng.normal(0, 0.01, obs_dim).astype(np.float32)
            w['b_var'][:] += rng.normal(0, 0.3, obs_dim).astype(np.float32)
    return pb


def with_action_bounds(pb, low, high):
    """The same problem with another action Box (MpcPolicy.sampling_params, mpc_policy.py:45-57): per-dimension (low, high), or a Box
    with an infinite bound anywhere — `is_bounded()` is then False and EVERY dimension takes the +-100 / 0 / 100 branch.  The
    normaliser's action columns are the Box when it is bounded and the range the samples then cover (+-100) when it is not — the data
    range TransitionModel._fit_statistics falls back to (transition_model.py:42-50) — so scaled inputs stay O(1)."""
    low, high = np.asarray(low, np.float32), np.asarray(high, np.float32)
    A = pb['low'].shape[0]
    assert low.shape == (A,) and high.shape == (A,)
    pb = dict(pb)
    pb['low'], pb['high'] = low, high
    O = pb['state'].shape[0]
    imin, imax = pb['inputs_min'].copy(), pb['inputs_max'].copy()
    bounded = bool(np.all(np.isfinite(low)) and np.all(np.isfinite(high)))
    imin[O:] = low if bounded else np.float32(-100.0)
    imax[O:] = high if bounded else np.float32(100.0)
    pb['inputs_min'], pb['inputs_max'] = imin, imax
    return pb


def random_action_bounds(rng, A):
    """(kind, low, high): 'unit' Box(-1, 1), 'asym' per-dimension bounds of mixed sign / width (one dimension may be a single point),
    'unbounded' (all infinite) or 'mixed' (some infinite: the reference then ignores the finite ones too)."""
    kind = str(rng.choice(['unit', 'asym', 'asym', 'asym', 'asym', 'unbounded', 'mixed', 'mixed']))
    if kind == 'unit':
        return kind, -np.ones(A, np.float32), np.ones(A, np.float32)
    centre = rng.uniform(-2.0, 2.0, A)
    half = rng.uniform(0.05, 1.5, A)
    if kind == 'asym' and A > 1 and rng.random() < 0.25:
        half[int(rng.integers(0, A))] = 0.0                 # low == high: sigma0 = 0, every sample of that dimension is the bound
    low, high = (centre - half).astype(np.float32), (centre + half).astype(np.float32)
    if kind == 'unbounded':
        low[:], high[:] = -np.inf, np.inf
    elif kind == 'mixed':
        j = int(rng.integers(0, A))
        if rng.random() < 0.5:
            high[j] = np.inf
        else:
            low[j] = -np.inf
    return kind, low, high


def configs(pb, N, H, P, E, k, I=3, variant='cem', thr=-1.0, noise=0.0, post=0.3, smoothing=0.0,
            sampling=True, scale=True, world_size=1, rank=0, chunks_per_tile=0, use_graph=False, rollout_segments=0, select_mode=0, precision='fp32'):
    sp = pb['scorer']
    O, A = pb['state'].shape[0], pb['low'].shape[0]
    # CEM_TEST_PRECISION=bf16x3: run every eligible case (units <= 128, relu) of the suites on the split-product rollout (a one-off coverage run)
    if os.environ.get('CEM_TEST_PRECISION') and pb['weights'][0]['W'][0].shape[1] <= 128 and pb['weights'][0].get('activation', 'relu') == 'relu':
        precision = os.environ['CEM_TEST_PRECISION']
    ocfg = o.PlanConfig(horizon=H, iterations=I, n_samples=N, n_elite=k, particles=P, ensemble_size=E, smoothing=smoothing,
                        stddev_threshold=thr, noise_stddev=noise, variant=variant, posterior_mean_threashold=post,
                        scale_features=scale, sampling_propagation=sampling)
    pcfg = PlannerConfig(obs_dim=O, act_dim=A, ensemble_size=E, particles=P, n_samples=N, horizon=H, n_elite=k, iterations=I,
                         scorer=ScorerConfig(goal_slice=sp.goal_slice, observe_goal_lidar=sp.observe_goal_lidar,
                                             lidar_max_dist=sp.lidar_max_dist, goal_size=sp.goal_size,
                                             reward_distance=sp.reward_distance, reward_goal=sp.reward_goal,
                                             reward_clip=sp.reward_clip, constrain_indicator=sp.constrain_indicator,
                                             cost_kinds=list(sp.cost_kinds)),
                         act_low=pb['low'], act_high=pb['high'], n_layers=len(pb['weights'][0]['W']), units=pb['weights'][0]['W'][0].shape[1],
                         activation=pb['weights'][0].get('activation', 'relu'),
                         smoothing=smoothing,
                         stddev_threshold=thr, noise_stddev=noise, variant=variant, posterior_mean_threashold=post,
                         sampling_propagation=sampling, scale_features=scale, world_size=world_size, rank=rank,
                         chunks_per_tile=chunks_per_tile, use_graph=use_graph, rollout_segments=rollout_segments, select_mode=select_mode,
                         precision=precision)
    return ocfg, pcfg


def noise(I, N, H, A, P, O, seed=0):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((I, N, H, A)).astype(np.float32),
            rng.standard_normal((I, H, P * N, O)).astype(np.float32),
            rng.standard_normal((A,)).astype(np.float32))


def make_planner(pb, pcfg):
    from ethz_safe_learning_amd import CemPlanner
    pl = CemPlanner(pcfg)
    pl.set_weights(pb['weights'])
    pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
    return pl


def elite_sets_equal_modulo_ties(scores, elite_a, elite_b, tol):
    """Two top-k sets agree if every candidate in the symmetric difference scores within tol of the k-th score."""
    a, b = set(int(x) for x in elite_a), set(int(x) for x in elite_b)
    if a == b:
        return True
    kth = np.sort(scores)[::-1][len(a) - 1]
    return all(abs(float(scores[i]) - float(kth)) <= tol for i in a ^ b)


# --------------------------------------------------------------------------------------------------------------------
# scorer branches of SafetyGymStateScorer (safety_gym.py:145-176): several constrained kinds, the non-indicator sum,
# observe_goal_dist instead of the goal lidar, no / active reward clip
# --------------------------------------------------------------------------------------------------------------------
# lidar-like slices per observation width: goal lidar, then the four cost kinds in the reference's order vases, hazards,
# pillars, gremlins (safety_gym.py:148-163), and the feature used as goal_dist.  obs 60 (obs+act <= 64: one input block per
# wave) reuses PointGoal1's layout and squeezes pillars / gremlins into the 3-wide gyro / velocimeter slots; obs 84 (two
# input blocks per wave) has room for two more 8-bin lidars.
SCORER_LAYOUTS = {
    60: dict(goal=(3, 19), kinds=[(41, 57), (22, 38), (19, 22), (57, 60)], goal_dist=0),
    84: dict(goal=(3, 19), kinds=[(41, 57), (22, 38), (64, 72), (72, 80)], goal_dist=60),
}
SCORER_SIZE_FRAC = 0.7
SCORER_CASES = {
    # name: (which of the four kinds are constrained, overrides)
    'two_kinds': dict(kinds=[0, 1]),
    'three_kinds': dict(kinds=[0, 1, 2]),
    'four_kinds': dict(kinds=[0, 1, 2, 3]),
    'four_kinds_sum': dict(kinds=[0, 1, 2, 3], constrain_indicator=False),
    'one_kind_sum': dict(kinds=[1], constrain_indicator=False),
    'no_cost_kinds': dict(kinds=[]),
    'goal_dist': dict(kinds=[1], observe_goal_lidar=False),                                  # observe_goal_dist (safety_gym.py:172-174)
    'goal_dist_sum3': dict(kinds=[0, 1, 2], observe_goal_lidar=False, constrain_indicator=False),
    'no_reward_clip': dict(kinds=[1], reward_clip=0.0, reward_goal=25.0),                    # falsy clip: no clipping (safety_gym.py:141)
    'active_reward_clip': dict(kinds=[1], reward_clip=0.02),                                 # |r| exceeds the clip on most steps
}


def scorer_problem(case, obs_dim=60, seed=31, E=5):
    """A problem whose scorer exercises the named branch.  Lidar-like slices get lidar-like state values (closest distances
    straddle the kinds' sizes) and the [0, 1] normaliser."""
    spec, lay = SCORER_CASES[case], SCORER_LAYOUTS[obs_dim]
    pb = make_problem(obs_dim, 2, E, 4, seed=seed)
    rng = np.random.default_rng(seed + 5)
    sp = pb['scorer']
    for lo, hi in [lay['goal']] + lay['kinds']:
        pb['state'][lo:hi] = rng.uniform(0.2, 0.9, hi - lo).astype(np.float32)
        pb['inputs_min'][lo:hi] = 0.0
        pb['inputs_max'][lo:hi] = 1.0
    goal_lidar = spec.get('observe_goal_lidar', True)
    goal_slice = lay['goal'] if goal_lidar else (lay['goal_dist'], lay['goal_dist'] + 1)
    if not goal_lidar:
        pb['state'][goal_slice[0]] = np.float32(0.27)          # goal_dist a little above 0.8 * goal_size: some rollouts reach the goal
    # a kind's size sits just under the closest distance of the real s_0, so predicted states drift in and out of it
    kinds = [(lay['kinds'][i][0], lay['kinds'][i][1],
              float(np.float32(SCORER_SIZE_FRAC * sp.lidar_max_dist * pb['state'][lay['kinds'][i][0]:lay['kinds'][i][1]].min()))) for i in spec['kinds']]
    pb['scorer'] = o.ScorerParams(goal_slice=goal_slice, observe_goal_lidar=goal_lidar,
                                  lidar_max_dist=sp.lidar_max_dist, goal_size=sp.goal_size, reward_distance=sp.reward_distance,
                                  reward_goal=spec.get('reward_goal', sp.reward_goal), reward_clip=spec.get('reward_clip', sp.reward_clip),
                                  constrain_indicator=spec.get('constrain_indicator', True), cost_kinds=kinds)
    return pb


# --------------------------------------------------------------------------------------------------------------------
# scores against the oracle, near-threshold candidates included
# --------------------------------------------------------------------------------------------------------------------
NEAR = 1e-4          # a `<=` comparison whose operands are closer than this may resolve either way within the rollout's rounding


def objective(traj, P, n, sp, variant, post):
    return o.compute_objective_safe(traj, P, n, sp, post) if variant == 'safe' else o.compute_objective_cem(traj, P, n, sp)


def admissible_scores(traj64, P, n, sp, variant, post, delta=2 * NEAR):
    """The oracle's objective under the ways a trajectory's near-threshold comparisons can legitimately resolve: the nominal
    thresholds, and the goal threshold / the cost sizes each moved down or up by `delta` (which flips exactly the comparisons whose
    operands lie within delta of the threshold) — nine score vectors, the nominal one first."""
    import dataclasses
    out = []
    for dg in (0, -1, 1):
        for dc in (0, -1, 1):
            v = dataclasses.replace(sp, goal_size=sp.goal_size + dg * delta / 0.8,
                                    cost_kinds=[(lo, hi, size + dc * delta) for lo, hi, size in sp.cost_kinds])
            out.append(objective(traj64, P, n, v, variant, post))
    return out


def assert_scores_match_oracle(scores_gpu, traj64, P, n, sp, variant, post, atol, what=''):
    """Every candidate's GPU score equals the fp64 oracle's within atol + half an fp32 ulp — candidates with a comparison within NEAR
    of its threshold (rounding may flip it) must equal the oracle's score under one of the admissible resolutions instead of being
    skipped.  Returns (max error over the clear candidates, number of near-threshold candidates, how many of those took a flipped
    outcome)."""
    outs = admissible_scores(traj64, P, n, sp, variant, post)
    ref = outs[0]
    allow = atol + 6e-8 * np.abs(ref)
    err = np.abs(scores_gpu - ref)
    near = o.threshold_margins(traj64, sp).reshape(P, n).min(axis=0) <= NEAR
    clear_bad = np.nonzero(~near & (err > allow))[0]
    assert clear_bad.size == 0, '%s: %d clear candidates differ from the oracle, e.g. %d: gpu %.9g oracle %.9g' % (
        what, clear_bad.size, clear_bad[0], scores_gpu[clear_bad[0]], ref[clear_bad[0]])
    best = np.min(np.stack([np.abs(scores_gpu - r) / (atol + 6e-8 * np.abs(r)) for r in outs]), axis=0)
    near_bad = np.nonzero(near & (best > 1.0))[0]
    assert near_bad.size == 0, '%s: %d near-threshold candidates equal none of the admissible outcomes, e.g. %d: gpu %.9g nominal %.9g' % (
        what, near_bad.size, near_bad[0], scores_gpu[near_bad[0]], ref[near_bad[0]])
    flipped = int((near & (err > allow)).sum())
    return (float(err[~near].max()) if (~near).any() else 0.0), int(near.sum()), flipped
