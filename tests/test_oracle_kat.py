"""Known-answer tests that pin oracle/cem_oracle.py to the reference's formulas.

The reference ships no tests or golden vectors: each
expected value below is derived by hand from the cited reference line.
"""
import numpy as np
import pytest

from oracle import cem_oracle as o


def test_softplus_three_branches():
    # Eigen softplus: x > 13.94 -> x ; x < -13.94 -> exp(x) ; else log1p(exp(x))   
    x = np.array([20.0, -20.0, 0.0, -8.0], np.float32)
    y = o.softplus_tf(x)
    assert y[0] == np.float32(20.0)
    assert y[1] == np.exp(np.float32(-20.0))
    assert abs(float(y[2]) - np.log(2.0)) < 1e-7
    assert abs(float(y[3]) - np.log1p(np.exp(-8.0))) < 1e-10


def test_activations_follow_tensorflow_definitions():
    """mlp_params['activation'] is eval'ed by the reference (mlp_ensemble.py:14): values and gradients of the TensorFlow functions
    at hand-picked points, incl. the kinks (relu' (0) = 0, leaky_relu' (0) = alpha = 0.2, elu' (0) = 1)."""
    z = np.array([-2.0, -0.5, 0.0, 0.5, 2.0])
    want = {'tf.nn.relu': ([0, 0, 0, 0.5, 2.0], [0, 0, 0, 1, 1]),
            'tf.nn.tanh': (np.tanh(z), 1 - np.tanh(z) ** 2),
            'tf.nn.sigmoid': (1 / (1 + np.exp(-z)), np.exp(-z) / (1 + np.exp(-z)) ** 2),
            'tf.nn.elu': ([np.exp(-2.0) - 1, np.exp(-0.5) - 1, 0, 0.5, 2.0], [np.exp(-2.0), np.exp(-0.5), 1, 1, 1]),
            'tf.nn.leaky_relu': ([-0.4, -0.1, 0, 0.5, 2.0], [0.2, 0.2, 0.2, 1, 1]),
            'tf.nn.softplus': (np.log1p(np.exp(z)), 1 / (1 + np.exp(-z))),
            # selu(z) = 1.0507009873554805 * (z, or 1.6732632423543772 * (e^z - 1) below zero); selu'(0) = scale (TensorFlow's SeluGrad tests out < 0)
            'tf.nn.selu': ([1.0507009873554805 * 1.6732632423543772 * (np.exp(-2.0) - 1), 1.0507009873554805 * 1.6732632423543772 * (np.exp(-0.5) - 1), 0,
                            1.0507009873554805 * 0.5, 1.0507009873554805 * 2.0],
                           [1.0507009873554805 * 1.6732632423543772 * np.exp(-2.0), 1.0507009873554805 * 1.6732632423543772 * np.exp(-0.5),
                            1.0507009873554805, 1.0507009873554805, 1.0507009873554805])}
    for name, (fv, dv) in want.items():
        f, df = o.activation_and_grad(name)
        np.testing.assert_allclose(f(z), fv, rtol=1e-12, atol=1e-15, err_msg=name)
        np.testing.assert_allclose(df(z), dv, rtol=1e-12, atol=1e-15, err_msg=name)
        assert o.activation_and_grad(name.split('.')[-1])[0](z).tolist() == f(z).tolist()      # bare names too
    # swish (= silu) and gelu (TensorFlow's default exact form): known values — swish(1) = 0.7310585786300049, swish'(0) = 0.5,
    # gelu(1) = Phi(1) = 0.8413447460685429, gelu'(0) = 0.5, gelu'(1) = Phi(1) + phi(1) = 1.0833154705876864 — and f' against a central difference
    f, df = o.activation_and_grad('tf.nn.swish')
    np.testing.assert_allclose(f(np.array([1.0, 0.0, -1.0])), [0.7310585786300049, 0.0, -0.2689414213699951], rtol=1e-14)
    np.testing.assert_allclose(df(np.array([0.0, 1.0])), [0.5, 0.9276705118714867], rtol=1e-14)
    assert o.activation_and_grad('tf.nn.silu')[0](z).tolist() == f(z).tolist()
    g, dg = o.activation_and_grad('tf.nn.gelu')
    np.testing.assert_allclose(g(np.array([1.0, 0.0, -1.0])), [0.8413447460685429, 0.0, -0.15865525393145707], rtol=1e-14)
    np.testing.assert_allclose(dg(np.array([0.0, 1.0])), [0.5, 1.0833154705876864], rtol=1e-14)
    for fn, dfn in ((f, df), (g, dg)):
        zz = np.linspace(-4.0, 4.0, 41)
        np.testing.assert_allclose(dfn(zz), (fn(zz + 1e-6) - fn(zz - 1e-6)) / 2e-6, rtol=1e-7, atol=1e-9)
    with pytest.raises(NotImplementedError):
        o.activation_and_grad('tf.nn.crelu')           # (changes the layer's width: not a pointwise activation)


def test_scale_rule():
    # transition_model.py:83-87: delta < 1e-5 -> 1.01
    x = np.array([[2.0, 5.0, 1.0]], np.float32)
    mn = np.array([0.0, 5.0, -1.0], np.float32)
    mx = np.array([4.0, 5.0, 1.0], np.float32)
    y = o.scale(x, mn, mx, True)
    np.testing.assert_allclose(y, [[0.5, 0.0 / 1.01, 1.0]], rtol=0, atol=1e-7)
    assert o.scale(x, mn, mx, False) is x


def _zero_weights(D, O, U=8, L=2, E=2):
    ws = []
    for _ in range(E):
        Ws, bs, fi = [], [], D
        for _ in range(L):
            Ws.append(np.zeros((fi, U), np.float32)); bs.append(np.zeros(U, np.float32)); fi = U
        ws.append(dict(W=Ws, b=bs, W_mu=np.zeros((U, O), np.float32), b_mu=np.zeros(O, np.float32),
                       W_var=np.zeros((U, O), np.float32), b_var=np.zeros(O, np.float32)))
    return ws


def test_zero_weight_net_gives_constant_trajectory_and_ln2_variance():
    # mlp_ensemble.py:30: var = softplus(0) + 1e-4 = ln 2 + 1e-4 ; mu = 0  => with sampling off s_t is constant
    O, A, B, H = 6, 2, 4, 3
    ws = _zero_weights(O + A, O)
    members = o.member_of_rows(B, 2)
    x = np.ones((B, O + A), np.float32)
    mu, var = o.ensemble_forward(x, ws, members)
    assert np.all(mu == 0)
    np.testing.assert_allclose(var, np.log(2.0) + 1e-4, rtol=1e-6)
    s0 = np.arange(B * O, dtype=np.float32).reshape(B, O)
    acts = np.zeros((B, H, A), np.float32)
    eps = np.ones((H, B, O), np.float32)
    mn, mx = np.zeros(O + A, np.float32), np.ones(O + A, np.float32)
    traj = o.unfold_sequences(s0, acts, ws, members, mn, mx, eps, True, sampling_propagation=False)
    for t in range(H + 1):
        np.testing.assert_array_equal(traj[:, t], s0)
    # with sampling: s_{t+1} = s_t + sqrt(var) * eps   (mlp_ensemble.py:192-193, transition_model.py:75)
    traj = o.unfold_sequences(s0, acts, ws, members, mn, mx, eps, True, sampling_propagation=True)
    sd = np.sqrt(np.float32(np.log(2.0) + 1e-4))
    np.testing.assert_allclose(traj[:, 2] - s0, 2 * sd, rtol=1e-5)


def test_row_to_member_map():
    # mlp_ensemble.py:123-126: tf.split into E contiguous chunks of B/E rows
    np.testing.assert_array_equal(o.member_of_rows(12, 3), [0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2])
    with pytest.raises(ValueError):
        o.member_of_rows(10, 3)
    # shipped cem_mpc config: E=15, P=5, N=150 -> chunks of 50 rows: candidates of one particle hit 3 members
    m = o.member_of_rows(750, 15)
    assert list(m[[0, 49, 50, 149, 150]]) == [0, 0, 1, 2, 3]


def test_single_dense_layer_identity():
    # Keras Dense = x @ W + b, W laid out [in, out] (mlp_ensemble.py:13)
    O, A = 3, 1
    W = np.zeros((4, 4), np.float32); W[0, 1] = 2.0      # out[1] = 2 * in[0]
    w = dict(W=[W], b=[np.array([0, 0, 0, -1], np.float32)], W_mu=np.eye(4, 3, dtype=np.float32),
             b_mu=np.array([0.5, 0, 0], np.float32), W_var=np.zeros((4, 3), np.float32), b_var=np.zeros(3, np.float32))
    x = np.array([[1.0, 2.0, 3.0, 4.0]], np.float32)
    mu, var = o.gaussian_dist_mlp(x, w)
    # h = relu([0, 2, 0, -1]) = [0, 2, 0, 0]; mu = h[:3] + b_mu
    np.testing.assert_allclose(mu, [[0.5, 2.0, 0.0]])


def test_beta_prior_and_threshold():
    # safe_cem_mpc.py:113-116 with mu=0.5, sigma=0.27: alpha = beta = ((0.5/0.0729) - 2) * 0.25 = 1.214678
    a, b = o.beta_prior(0.5, 0.27, np.float64)
    assert abs(a - 1.2146776406035664) < 1e-12 and abs(b - a) < 1e-12
    # P = 45, threshold 0.15 (config/policies.yaml:17,20): safe iff count <= 5
    P, N = 45, 8
    for count, expect in [(0, True), (5, True), (6, False), (45, False)]:
        c = np.zeros((P, N), np.float32); c[:count, :] = 1.0
        safe = o.bayesian_safety_beta_inference(c.reshape(-1), P, N, 0.15)
        assert bool(safe.all()) == expect and bool(safe.any()) == expect
    # with P = 5 nothing is ever safe at 0.15: (1.2147 + 0) / (2.4294 + 5) = 0.1635 > 0.15
    assert not o.bayesian_safety_beta_inference(np.zeros(5 * 3, np.float32), 5, 3, 0.15).any()


def test_closest_distance_and_clip():
    # safety_gym.py:188-192: min_bins clip(D - D*(1-x), 0, D)
    sp = o.ScorerParams(goal_slice=(0, 2), lidar_max_dist=4.0)
    lid = np.array([[0.5, 0.25], [1.5, 2.0], [-0.1, 0.9]], np.float32)
    np.testing.assert_allclose(o.closest_distance(lid, sp), [1.0, 4.0, 0.0])


def test_reward_and_goal_bonus():
    # safety_gym.py:113-119: r = (d - d')*reward_distance + 1[d <= 0.8*goal_size]*reward_goal, clipped
    sp = o.ScorerParams(goal_slice=(0, 1), lidar_max_dist=4.0, goal_size=0.3, reward_distance=1.0, reward_goal=1.0, reward_clip=10.0)
    obs = np.array([[0.25], [0.055], [0.05]], np.float32)         # d = 1.0, 0.22, 0.2
    nxt = np.array([[0.125], [0.055], [0.0]], np.float32)         # d' = 0.5, 0.22, 0.0
    r, ga = o.reward(obs, nxt, sp)
    np.testing.assert_allclose(r, [0.5, 1.0, 1.2], rtol=1e-6)
    np.testing.assert_array_equal(ga, [False, True, True])        # d <= fl32(0.3 * 0.8)
    # exactly on the boundary the reference's formula order decides: 4 - 4*(1 - fl32(0.06)) = 0.24000001 > fl32(0.24)
    assert not o.reward(np.array([[0.06]], np.float32), np.array([[0.06]], np.float32), sp)[1][0]
    sp.reward_clip = 0.3
    r, _ = o.reward(obs, nxt, sp)
    np.testing.assert_allclose(r, [0.3, 0.3, 0.3], rtol=1e-6)


def test_cost_kinds_and_indicator():
    # safety_gym.py:145-166
    sp = o.ScorerParams(goal_slice=(0, 1), lidar_max_dist=4.0, constrain_indicator=False,
                        cost_kinds=[(1, 2, 0.2), (2, 3, 0.4)])
    obs = np.array([[0, 0.04, 0.09], [0, 0.04, 0.5], [0, 0.5, 0.5]], np.float32)   # dists (.16,.36) (.16,2) (2,2)
    np.testing.assert_allclose(o.cost(obs, sp), [2.0, 1.0, 0.0])
    sp.constrain_indicator = True
    np.testing.assert_allclose(o.cost(obs, sp), [1.0, 1.0, 0.0])


def _traj_from_goal_dists(dists, hazard=None):
    """rows x (H+1) goal distances -> traj with obs = [goal_lidar(1 bin), hazard_lidar(1 bin)], D = 4."""
    d = np.asarray(dists, np.float32)
    traj = np.zeros(d.shape + (2,), np.float32)
    traj[..., 0] = d / 4.0
    traj[..., 1] = 1.0 if hazard is None else np.asarray(hazard, np.float32) / 4.0
    return traj


def test_done_masking_differs_between_cem_and_safe():
    # one row: d = 1.0 -> 0.2 -> 0.1 -> 0.05.  goal_achieved(s_t) at t=1,2 (d <= 0.24).
    sp = o.ScorerParams(goal_slice=(0, 1), lidar_max_dist=4.0, goal_size=0.3, reward_clip=10.0, cost_kinds=[(1, 2, 0.2)])
    traj = _traj_from_goal_dists([[1.0, 0.2, 0.1, 0.05]])
    # mpc_policy.py:34-37: reward of the step where s_t is first at goal IS counted: 0.8 + (0.1 + 1) = 1.9
    np.testing.assert_allclose(o.compute_objective_cem(traj, 1, 1, sp), [1.9], rtol=1e-6)
    # safe_cem_mpc.py:87-93: done OR-ed first, so that step is masked: 0.8 (minus 100: P=1 is never safe at 0.15)
    np.testing.assert_allclose(o.compute_objective_safe(traj, 1, 1, sp, 0.15), [0.8 - 100.0], rtol=1e-6)
    np.testing.assert_allclose(o.compute_objective_safe(traj, 1, 1, sp, 0.9), [0.8], rtol=1e-6)


def test_safe_objective_counts_costs_per_step_over_particles():
    # P=3 particles, N=2 candidates, H=2.  Candidate 0: two particles in a hazard at t=0; candidate 1: none.
    sp = o.ScorerParams(goal_slice=(0, 1), lidar_max_dist=4.0, goal_size=0.3, reward_clip=10.0, cost_kinds=[(1, 2, 0.2)])
    d = np.full((6, 3), 2.0, np.float32)
    hz = np.full((6, 3), 3.0, np.float32)
    hz[0, 0] = 0.1; hz[2, 0] = 0.1            # rows p*N+n: (p=0,n=0) and (p=1,n=0)
    traj = _traj_from_goal_dists(d, hz)
    # posterior(count=2) = (1.2147+2)/(2.4294+3) = 0.592 ; posterior(0) = 0.2237
    s = o.compute_objective_safe(traj, 3, 2, sp, 0.3)
    np.testing.assert_allclose(s, [-100.0, 0.0], atol=1e-6)
    s = o.compute_objective_safe(traj, 3, 2, sp, 0.6)
    np.testing.assert_allclose(s, [0.0, 0.0], atol=1e-6)


def test_top_k_ties_prefer_lower_index():
    np.testing.assert_array_equal(o.top_k(np.array([1, 3, 3, 2, 3], np.float32), 2), [1, 2])
    np.testing.assert_array_equal(o.top_k(np.array([5, 1, 4], np.float32), 3), [0, 1, 2])
    assert o.best_of_elite(np.array([1, 3, 3, 2, 3], np.float32), np.array([1, 2])) == 1


def test_moments_are_population():
    x = np.array([[1.0, 10.0], [3.0, 10.0], [5.0, 16.0]], np.float32)
    mean, var = o.moments(x)
    np.testing.assert_allclose(mean, [3.0, 12.0])
    np.testing.assert_allclose(var, [8.0 / 3.0, 8.0])


def test_sampling_params():
    lb, ub, mu, sg = o.sampling_params([-1, 0], [1, 4])            # mpc_policy.py:47-51
    np.testing.assert_allclose(mu, [0, 2]); np.testing.assert_allclose(sg, [1, 2])
    lb, ub, mu, sg = o.sampling_params([-np.inf, 0], [np.inf, 4])  # mpc_policy.py:53-56
    np.testing.assert_allclose(lb, [-100, -100]); np.testing.assert_allclose(sg, [100, 100])


def _tiny(seed=0, variant='cem', thr=-1.0, iters=3):
    pb = o.synthetic_problem(obs_dim=6, act_dim=2, ensemble_size=2, units=8, n_layers=2, seed=seed)
    cfg = o.PlanConfig(horizon=3, iterations=iters, n_samples=8, n_elite=3, particles=2, ensemble_size=2,
                       stddev_threshold=thr, noise_stddev=0.1, variant=variant, posterior_mean_threashold=0.4)
    rng = np.random.default_rng(seed + 100)
    ea = rng.standard_normal((iters, 8, 3, 2)).astype(np.float32)
    em = rng.standard_normal((iters, 3, 16, 6)).astype(np.float32)
    eo = rng.standard_normal(2).astype(np.float32)
    return pb, cfg, ea, em, eo


def test_plan_returns_best_sampled_first_action_plus_noise():
    pb, cfg, ea, em, eo = _tiny()
    tr = []
    a, s, it = o.do_generate_action(pb['state'], pb['weights'], pb['inputs_min'], pb['inputs_max'], pb['low'], pb['high'],
                                    ea, em, eo, cfg, pb['scorer'], trace=tr)
    assert it == 3
    # best-so-far is the first action of the best candidate ever seen (cem_mpc.py:57-60), not mu[0]
    best_it = int(np.argmax([t['scores'].max() for t in tr]))
    j = int(np.argmax(tr[best_it]['scores']))
    np.testing.assert_allclose(a, tr[best_it]['actions'][j, 0] + eo * np.float32(0.1), rtol=1e-6)
    assert s == tr[best_it]['scores'].max()
    # actions are clipped to the box (cem_mpc.py:48)
    assert all(np.all(np.abs(t['actions']) <= 1.0) for t in tr)
    # refit: mu, sigma are the elite moments with smoothing 0 (cem_mpc.py:61-65)
    m, v = o.moments(tr[0]['actions'][tr[0]['elite']])
    np.testing.assert_allclose(tr[0]['mu'], m, rtol=1e-6)
    np.testing.assert_allclose(tr[0]['sigma'], np.sqrt(v), rtol=1e-6)


def test_smoothing_factors_are_each_rounded_once():
    """cem_mpc.py:64-65: `self.smoothing * mu + (1.0 - self.smoothing) * mean` — both factors are Python floats that TF converts
    once to fp32, so the second one is fl32(1.0 - s) evaluated in double.  At s = 0.09 that differs by one ulp from
    1.0f - fl32(s), the form rounds 1 and 2 of this repo used on both sides."""
    s = 0.09
    once = np.float32(1.0 - s)
    twice = np.float32(1.0) - np.float32(s)
    assert once != twice and abs(float(once) - float(twice)) < 1e-7          # the case is a discriminating one
    n_diff = sum(np.float32(1.0 - i / 100.0) != np.float32(1.0) - np.float32(i / 100.0) for i in range(1, 100))
    assert n_diff == 41
    rng = np.random.default_rng(5)
    N, H, A, k = 8, 3, 2, 1
    actions = rng.uniform(-1, 1, (N, H, A)).astype(np.float32)
    scores = np.arange(N, dtype=np.float32)                                  # elite = the last candidate alone: var = 0 exactly
    mu0 = rng.uniform(-1, 1, (H, A)).astype(np.float32)
    sg0 = rng.uniform(0.5, 1, (H, A)).astype(np.float32)
    cfg = o.PlanConfig(horizon=H, iterations=1, n_samples=N, n_elite=k, particles=1, ensemble_size=1, smoothing=s)
    mu, sigma, *_ = o.select_and_refit(scores, actions, mu0, sg0, np.zeros(A, np.float32), np.float32(-np.inf), cfg)
    np.testing.assert_array_equal(mu, np.float32(s) * mu0 + once * actions[N - 1])
    np.testing.assert_array_equal(sigma, np.float32(s) * sg0 + once * np.zeros((H, A), np.float32))
    assert np.any(mu != np.float32(s) * mu0 + twice * actions[N - 1])


def test_early_stop_runs_at_least_one_iteration():
    pb, cfg, ea, em, eo = _tiny(thr=10.0)              # mean(sigma) <= 10 after the first refit (cem_mpc.py:66-67)
    a, s, it = o.do_generate_action(pb['state'], pb['weights'], pb['inputs_min'], pb['inputs_max'], pb['low'], pb['high'],
                                    ea, em, eo, cfg, pb['scorer'])
    assert it == 1


def test_fp32_tracks_fp64_shadow():
    pb, cfg, ea, em, eo = _tiny(seed=3, variant='safe')
    t32, t64 = [], []
    o.do_generate_action(pb['state'], pb['weights'], pb['inputs_min'], pb['inputs_max'], pb['low'], pb['high'],
                         ea, em, eo, cfg, pb['scorer'], trace=t32)
    o.do_generate_action(pb['state'], o.cast_weights(pb['weights'], np.float64), pb['inputs_min'], pb['inputs_max'],
                         pb['low'], pb['high'], ea, em, eo, cfg, pb['scorer'], dtype=np.float64, trace=t64)
    np.testing.assert_allclose(t32[0]['scores'], t64[0]['scores'], rtol=1e-5, atol=1e-5)


def test_sharded_candidate_scores_equal_full_batch():
    """Evaluating a candidate shard with the members its GLOBAL rows select reproduces the full-batch scores
    (the property the multi-GPU partitioning relies on)."""
    pb, cfg, ea, em, eo = _tiny(seed=5)
    lb, ub, mu0, sg0 = o.sampling_params(pb['low'], pb['high'])
    acts = o.sample_actions(np.broadcast_to(mu0, (3, 2)), np.broadcast_to(sg0, (3, 2)), lb, ub, ea[0])
    full = o.candidate_scores(pb['state'], acts, pb['weights'], pb['inputs_min'], pb['inputs_max'], em[0], cfg, pb['scorer'])
    N, P, W = 8, 2, 2
    parts = []
    for g in range(W):
        n0, n1 = g * N // W, (g + 1) * N // W
        rows = np.concatenate([p * N + np.arange(n0, n1) for p in range(P)])
        members = o.member_of_rows(P * N, cfg.ensemble_size, rows)
        parts.append(o.candidate_scores(pb['state'], acts[n0:n1], pb['weights'], pb['inputs_min'], pb['inputs_max'],
                                        em[0][:, rows], cfg, pb['scorer'], members=members))
    np.testing.assert_allclose(np.concatenate(parts), full, rtol=1e-6, atol=1e-6)
