"""The drop-in boundary: simba-shaped classes (same names / ctor kwargs as reference simba/policies, simba/models,
SURVEY 8b) constructed the way MbrlAgent does (mbrl_agent.py:103-118), at the reference's shipped hyper-parameters
(config/policies.yaml, config/models.yaml)."""
import numpy as np
import pytest

from oracle import cem_oracle as o

from ethz_safe_learning_amd.simba.environment_utils import SafetyGymStateScorer, SyntheticSafetyGym
from ethz_safe_learning_amd.simba.infrastructure.common import standardize_name
from ethz_safe_learning_amd.simba.models import MlpEnsemble, TransitionModel          # noqa: F401
from ethz_safe_learning_amd.simba.policies import CemMpc, PolicyBase, RandomMpc, SafeCemMpc   # noqa: F401
from ethz_safe_learning_amd.simba.spaces import Box

POLICIES_YAML = dict(                                    # reference config/policies.yaml:2-20
    cem_mpc=dict(horizon=8, iterations=10, smoothing=0.0, n_samples=150, n_elite=15, particles=5, stddev_threshold=0.25,
                 noise_stddev=0.001),
    safe_cem_mpc=dict(horizon=8, iterations=9, smoothing=0.0, n_samples=500, n_elite=20, particles=45, stddev_threshold=0.25,
                      noise_stddev=0.01, posterior_mean_threashold=0.15))
MODELS_YAML = dict(ensemble_size=15, batch_size=64, validation_split=0.2, learning_rate=0.00025, learning_rate_schedule=True,
                   training_steps=5000, mlp_params=dict(n_layers=4, units=128, activation='tf.nn.relu', dropout_rate=0.0))


def make_agent_parts(policy_name, seed=0, units=None, activation=None, dropout_rate=None, precision=None):
    """What MbrlAgent.__init__ does (mbrl_agent.py:27-35,103-118; agent_factory.py:22 injects train_epochs)."""
    env = SyntheticSafetyGym()
    model_params = dict(MODELS_YAML, scale_features=True, train_epochs=10, seed=seed)
    if units is not None:                                  # models.yaml:11 takes any width
        model_params['mlp_params'] = dict(MODELS_YAML['mlp_params'], units=units)
    if activation is not None:                             # models.yaml:12: any string the reference can eval
        model_params['mlp_params'] = dict(model_params['mlp_params'], activation=activation)
    if dropout_rate is not None:                           # models.yaml:13
        model_params['mlp_params'] = dict(model_params['mlp_params'], dropout_rate=dropout_rate)
    model = TransitionModel(model='mlp_ensemble', observation_space=env.observation_space, action_space=env.action_space,
                            sampling_propagation=True, **model_params)
    policy_params = dict(POLICIES_YAML[policy_name])
    policy_params['environment'] = env
    if precision is not None:
        policy_params['precision'] = precision
    policy = eval(standardize_name(policy_name))(model=model, **policy_params)
    return env, model, policy


def trained_like(model, rng):
    """Stand-in for MlpEnsemble.fit: shrink the heads and give the variance head a negative bias, then fit statistics."""
    ws = model.model.get_weights()
    for w in ws:
        w['W_mu'] *= 0.05; w['W_var'] *= 0.05; w['b_var'][:] = -8.0
    model.model.set_weights(ws)
    data = np.concatenate([rng.normal(0, 1, (256, model.observation_space_dim)), rng.uniform(-1, 1, (256, model.action_space_dim))], 1)
    model._fit_statistics(data.astype(np.float32))


def test_class_lookup_and_constructor_contract():
    assert standardize_name('safe_cem_mpc') == 'SafeCemMpc' and standardize_name('cem_mpc') == 'CemMpc'
    env, model, pol = make_agent_parts('safe_cem_mpc')
    assert isinstance(pol, SafeCemMpc) and isinstance(pol, CemMpc) and isinstance(pol, PolicyBase)
    assert pol.posterior_mean_threashold == 0.15 and pol.elite == 20 and pol.particles == 45
    cfg = pol.planner_config()
    assert (cfg.variant, cfg.ensemble_size, cfg.particles, cfg.n_samples, cfg.horizon, cfg.iterations) == ('safe', 15, 45, 500, 8, 9)
    assert cfg.scorer.goal_slice == (3, 19) and cfg.scorer.cost_kinds == [(22, 38, 0.2)]     # PointGoal1 sorted-key layout
    lb, ub, mu, sg = pol.sampling_params                                                        # mpc_policy.py:45-57
    np.testing.assert_array_equal(mu, [0, 0]); np.testing.assert_array_equal(sg, [1, 1])
    env, model, pol = make_agent_parts('cem_mpc')
    assert pol.planner_config().variant == 'cem'
    assert RandomMpc(env.action_space).generate_action(None).shape == (2,)
    # `random_shooting_mpc` is a key of the reference's policies.yaml whose class cannot be constructed there either
    from ethz_safe_learning_amd.simba.agents.mbrl_agent import _POLICIES
    with pytest.raises(NotImplementedError, match='random_shooting_mpc'):
        _POLICIES[standardize_name('random_shooting_mpc')](model=model, environment=env, horizon=8, n_samples=10, particles=5)


def test_transition_model_statistics_and_scale():
    env, model, _ = make_agent_parts('cem_mpc')
    assert np.isinf(model.inputs_min[0]) and model.inputs_min[3] == 0.0 and model.inputs_max[3] == 1.0   # :28-29
    rng = np.random.default_rng(0)
    data = rng.normal(0, 1, (100, 62)).astype(np.float32)
    v0 = model.version
    model._fit_statistics(data)                                                                 # :42-50
    assert model.version != v0
    assert model.inputs_min[0] == data[:, 0].min() and model.inputs_min[3] == 0.0 and model.inputs_max[61] == 1.0
    np.testing.assert_allclose(model.scale(data), o.scale(data, model.inputs_min, model.inputs_max, True), rtol=1e-6)
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):                 # training runs on the GPU only; no silent CPU fallback
            model.fit(data, data[:, :60])


def test_model_uid_is_unique_per_instance():
    """The planner-handle cache tags staged weights with (model.uid, model.version).  id() is recycled after garbage
    collection, so two models that never coexist can share an id — a uid never repeats."""
    import gc
    uids, ids = set(), set()
    for seed in range(6):
        env, model, _ = make_agent_parts('cem_mpc', seed=seed)
        assert model.uid not in uids
        uids.add(model.uid); ids.add(id(model))
        del env, model, _
        gc.collect()
    assert len(uids) == 6


def test_scorer_config_variants():
    table = dict(goal_dist=slice(0, 1), hazards_lidar=slice(1, 6), vases_lidar=slice(6, 11))
    s = SafetyGymStateScorer(dict(task='goal', observe_goal_lidar=False, observe_goal_dist=True, goal_size=0.3, lidar_max_dist=4,
                                  constrain_hazards=True, constrain_vases=True, hazards_size=0.2, vases_size=0.1,
                                  constrain_indicator=False, reward_distance=1.0, reward_goal=1.0, reward_clip=0), table)
    c = s.to_scorer_config()
    assert not c.observe_goal_lidar and c.goal_slice == (0, 1) and c.reward_clip == 0.0 and not c.constrain_indicator
    assert c.cost_kinds == [(6, 11, 0.1), (1, 6, 0.2)]                     # vases before hazards (safety_gym.py:148-156)
    with pytest.raises(NotImplementedError):
        SafetyGymStateScorer(dict(task='push'), table).to_scorer_config()
    import torch
    if not torch.cuda.is_available():                                      # no host scorer: reward / cost are HIP ops and fail loudly
        with pytest.raises(RuntimeError, match='no CPU'):
            s.reward(np.zeros((2, 11), np.float32), np.zeros((2, 11), np.float32))


def test_box():
    b = Box(-np.ones(3), np.ones(3))
    assert b.is_bounded() and b.shape == (3,) and b.sample().shape == (3,)
    assert not Box([-np.inf], [np.inf]).is_bounded()


@pytest.mark.gpu
@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
@pytest.mark.parametrize('policy_name', ['cem_mpc', 'safe_cem_mpc'])
def test_generate_action_at_shipped_config_matches_oracle(policy_name, precision):
    """policy.generate_action contract (agent.py:120,146) + parity at the reference's own hyper-parameters:
    E=15 with P=5,N=150 (candidates of one particle hit three members) and P=45,N=500 (three particles per member) — on the fp32
    kernels and on the split-product rollout (the policy kwarg `precision`), same oracle, same tolerances."""
    env, model, pol = make_agent_parts(policy_name, seed=3, precision=None if precision == 'fp32' else precision)
    trained_like(model, np.random.default_rng(1))
    pp = POLICIES_YAML[policy_name]
    I, N, H, P = pp['iterations'], pp['n_samples'], pp['horizon'], pp['particles']
    rng = np.random.default_rng(2)
    state = np.zeros(60, np.float64)                      # the env hands float64 observations (cast at cem_mpc.py:32)
    state[:] = rng.normal(0, 0.3, 60)
    for lo, hi in ((3, 19), (22, 38), (41, 57)):
        state[lo:hi] = rng.uniform(0.3, 0.9, hi - lo)
    a = pol.generate_action(state)
    assert a.shape == env.action_space.shape and a.dtype == np.float32 and np.all(np.isfinite(a))
    assert 1 <= pol.last_iterations <= I
    # parity on explicit noise
    ea = rng.standard_normal((I, N, H, 2)).astype(np.float32)
    em = rng.standard_normal((I, H, P * N, 60)).astype(np.float32)
    eo = rng.standard_normal(2).astype(np.float32)
    a, s = pol.do_generate_action(state, eps_act=ea, eps_model=em, eps_out=eo)
    cfg = pol.planner_config()
    ocfg = o.PlanConfig(horizon=H, iterations=I, n_samples=N, n_elite=pp['n_elite'], particles=P, ensemble_size=15,
                        smoothing=pp['smoothing'], stddev_threshold=pp['stddev_threshold'], noise_stddev=pp['noise_stddev'],
                        variant=cfg.variant, posterior_mean_threashold=pp.get('posterior_mean_threashold', 0.15))
    sp = o.ScorerParams(goal_slice=(3, 19), cost_kinds=[(22, 38, 0.2)])
    tr = []
    ra, rs, rit = o.do_generate_action(state.astype(np.float32), model.model.get_weights(), model.inputs_min, model.inputs_max,
                                       env.action_space.low, env.action_space.high, ea, em, eo, ocfg, sp, trace=tr)
    assert pol.last_iterations == rit, 'early stop (stddev_threshold 0.25) must trigger at the same iteration'
    assert abs(s - rs) <= 2e-5
    np.testing.assert_allclose(a, ra, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_same_shape_policies_restage_their_own_model():
    """Two policies of one shape share a cached planner handle; each must plan with ITS model's weights, also when the
    first model has been garbage collected (its id() may be handed to the second)."""
    import gc
    state = np.zeros(60, np.float64); state[3:19] = 0.5; state[22:38] = 0.6
    rng = np.random.default_rng(0)
    pp = POLICIES_YAML['cem_mpc']
    ea = rng.standard_normal((pp['iterations'], pp['n_samples'], pp['horizon'], 2)).astype(np.float32)
    scores = []
    for seed in (11, 12, 11):
        env, model, pol = make_agent_parts('cem_mpc', seed=seed)
        trained_like(model, np.random.default_rng(seed))
        I, N, H = pol.iterations, pol.n_samples, pol.horizon
        a, s = pol.do_generate_action(state, eps_act=ea, eps_model=np.zeros((I, H, pol.particles * N, 60), np.float32),
                                      eps_out=np.zeros(2, np.float32))
        scores.append(float(s))
        del env, model, pol
        gc.collect()
    assert scores[0] == scores[2] and scores[0] != scores[1], scores


@pytest.mark.gpu
def test_transition_model_unfold_api():
    env, model, _ = make_agent_parts('cem_mpc', seed=5)
    trained_like(model, np.random.default_rng(1))
    rng = np.random.default_rng(3)
    B, H = 30, 4
    s0 = rng.normal(0, 0.3, (B, 60)).astype(np.float32)
    acts = rng.uniform(-1, 1, (B, H, 2)).astype(np.float32)
    eps = rng.standard_normal((H, B, 60)).astype(np.float32)
    traj = model.simulate_trajectories(s0, acts, eps_model=eps)
    assert traj.shape == (B, H + 1, 60)
    ref = o.unfold_sequences(s0.astype(np.float64), acts.astype(np.float64), o.cast_weights(model.model.get_weights(), np.float64),
                             o.member_of_rows(B, 15), model.inputs_min, model.inputs_max, eps.astype(np.float64), True, True)
    assert np.abs(traj - ref).max() <= 5e-5
    pred = model.predict(np.concatenate([s0, acts[:, 0]], 1), eps_model=eps[:1])          # transition_model.py:51-55
    np.testing.assert_allclose(pred[:, 1], ref[:, 1], atol=5e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('policy_name', ['cem_mpc', 'safe_cem_mpc'])
def test_compute_objective_and_env_scorer_are_callable(policy_name):
    """MpcPolicy.compute_objective (mpc_policy.py:26-39 / safe_cem_mpc.py:76-96) and env.get_reward / get_cost
    (safety_gym.py:62-66) as the methods the reference exposes: numpy in, numpy out, against the oracle."""
    env, model, pol = make_agent_parts(policy_name, seed=4)
    trained_like(model, np.random.default_rng(1))
    pp = POLICIES_YAML[policy_name]
    P, n, H = pp['particles'], 12, 6
    rng = np.random.default_rng(7)
    traj = rng.uniform(-0.1, 1.0, (P * n, H + 1, 60)).astype(np.float32)
    traj[:, :, 22:38] = rng.uniform(0.0, 0.3, (P * n, H + 1, 16))             # hazards lidar near its 0.2 / 4 = 0.05 threshold
    traj[: P * n // 2, :, 22:38] += 0.2                                        # half the rows stay clear of the hazards
    scores = pol.compute_objective(traj, None)
    assert isinstance(scores, np.ndarray) and scores.shape == (n,) and scores.dtype == np.float32
    sp = o.ScorerParams(goal_slice=(3, 19), cost_kinds=[(22, 38, 0.2)])
    t64 = traj.astype(np.float64)
    ref = (o.compute_objective_safe(t64, P, n, sp, pp['posterior_mean_threashold']) if policy_name == 'safe_cem_mpc'
           else o.compute_objective_cem(t64, P, n, sp))
    ok = o.threshold_margins(t64, sp).reshape(P, n).min(axis=0) > 1e-5
    assert ok.sum() >= n // 2
    np.testing.assert_allclose(scores[ok], ref[ok], rtol=1e-6, atol=5e-6)
    # the environment adapters: actions are ignored, cost is evaluated on `obs` (safety_gym.py:62-66)
    obs, nxt = traj[:, 0], traj[:, 1]
    r, done = env.get_reward(obs, None, nxt)
    c = env.get_cost(obs, None, nxt)
    rr, dd = o.reward(obs, nxt, sp)
    np.testing.assert_array_equal(r, rr)
    np.testing.assert_array_equal(done, dd)
    np.testing.assert_array_equal(c, o.cost(obs, sp))


@pytest.mark.gpu
def test_predict_draws_fresh_noise_per_call():
    """Normal.sample() is fresh on every call in the reference (mlp_ensemble.py:189-193): repeated predict() calls on one
    input must differ, and pinning (seed, call) must reproduce."""
    env, model, _ = make_agent_parts('cem_mpc', seed=6)
    trained_like(model, np.random.default_rng(1))
    ws = model.model.get_weights()
    for w in ws:
        w['b_var'][:] = -2.0                                                    # a visible predictive spread
    model.model.set_weights(ws)
    x = np.random.default_rng(2).normal(0, 0.3, (30, 62)).astype(np.float32)
    a, b = model.predict(x), model.predict(x)
    assert not np.array_equal(a, b)
    np.testing.assert_array_equal(model.predict(x, seed=3, call=9), model.predict(x, seed=3, call=9))


def test_unknown_activation_is_refused_with_a_reason():
    with pytest.raises(NotImplementedError, match='crelu'):
        make_agent_parts('cem_mpc', activation='tf.nn.crelu')


@pytest.mark.gpu
@pytest.mark.parametrize('activation', ['tf.nn.elu', 'tf.nn.swish'])
def test_model_with_another_activation_fits_and_plans(activation):
    """models.yaml `activation: tf.nn.elu` (or tf.nn.swish — the usual choice of ensemble-dynamics papers, whose backward pass needs the kept
    pre-activations) through the simba classes: MlpEnsemble.fit on the device (the GEMM trainer), then
    SafeCemMpc.generate_action (the generic rollout kernel) against the oracle on the FITTED weights with that activation;
    `dropout_rate: 0.1` rides along (training-time only: the planner's forward passes run with training=False, mlp_ensemble.py:127)."""
    env, model, pol = make_agent_parts('safe_cem_mpc', seed=5, activation=activation, dropout_rate=0.1)   # Dropout acts in fit only
    assert model.model.dropout_rate == 0.1
    rng = np.random.default_rng(4)
    obs = rng.normal(0, 0.5, (400, 60)).astype(np.float32)
    acs = rng.uniform(-1, 1, (400, 2)).astype(np.float32)
    nxt = (obs + 0.05 * np.tanh(obs) + 0.02 * rng.normal(0, 1, obs.shape)).astype(np.float32)
    model.model.training_steps, model.model.train_epochs = 40, 1
    w0 = model.model.get_weights()[0]['W'][0].copy()
    model.fit(np.concatenate([obs, acs], axis=1), nxt)      # TransitionModel.fit(inputs = [obs | acs], targets = next_obs), transition_model.py:31-40
    ws = model.model.get_weights()
    assert np.abs(ws[0]['W'][0] - w0).max() > 1e-4, 'fit did not move the weights'
    pp = POLICIES_YAML['safe_cem_mpc']
    I, N, H, P = pp['iterations'], pp['n_samples'], pp['horizon'], pp['particles']
    state = rng.normal(0, 0.3, 60)
    for lo, hi in ((3, 19), (22, 38), (41, 57)):
        state[lo:hi] = rng.uniform(0.3, 0.9, hi - lo)
    ea = rng.standard_normal((I, N, H, 2)).astype(np.float32)
    em = rng.standard_normal((I, H, P * N, 60)).astype(np.float32)
    eo = rng.standard_normal(2).astype(np.float32)
    a, s = pol.do_generate_action(state, eps_act=ea, eps_model=em, eps_out=eo)
    for w in ws:
        w['activation'] = activation
    ocfg = o.PlanConfig(horizon=H, iterations=I, n_samples=N, n_elite=pp['n_elite'], particles=P, ensemble_size=15,
                        smoothing=pp['smoothing'], stddev_threshold=pp['stddev_threshold'], noise_stddev=pp['noise_stddev'],
                        variant='safe', posterior_mean_threashold=pp.get('posterior_mean_threashold', 0.15))
    ra, rs, rit = o.do_generate_action(state.astype(np.float32), ws, model.inputs_min, model.inputs_max, env.action_space.low,
                                       env.action_space.high, ea, em, eo, ocfg, o.ScorerParams(goal_slice=(3, 19), cost_kinds=[(22, 38, 0.2)]))
    assert pol.last_iterations == rit and abs(s - rs) <= 2e-5
    np.testing.assert_allclose(a, ra, rtol=1e-5, atol=1e-6)
