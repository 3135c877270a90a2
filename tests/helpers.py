"""Shared builders for the parity tests: one synthetic problem -> (oracle config, planner config)."""
import numpy as np

from oracle import cem_oracle as o

from ethz_safe_learning_amd import PlannerConfig, ScorerConfig


def make_problem(obs_dim=60, act_dim=2, E=5, n_layers=4, seed=1234, bias_noise=0.05, **kw):
    pb = o.synthetic_problem(obs_dim=obs_dim, act_dim=act_dim, ensemble_size=E, units=128, n_layers=n_layers, seed=seed, **kw)
    if bias_noise:
        rng = np.random.default_rng(seed + 999)
        for w in pb['weights']:
            for b in w['b']:
                b[:] = rng.normal(0, bias_noise, b.shape).astype(np.float32)
            w['b_mu'][:] = rng.normal(0, 0.01, obs_dim).astype(np.float32)
            w['b_var'][:] += rng.normal(0, 0.3, obs_dim).astype(np.float32)
    return pb


def configs(pb, N, H, P, E, k, I=3, variant='cem', thr=-1.0, noise=0.0, post=0.3, smoothing=0.0,
            sampling=True, scale=True, world_size=1, rank=0, chunks_per_tile=0, use_graph=False):
    sp = pb['scorer']
    O, A = pb['state'].shape[0], pb['low'].shape[0]
    ocfg = o.PlanConfig(horizon=H, iterations=I, n_samples=N, n_elite=k, particles=P, ensemble_size=E, smoothing=smoothing,
                        stddev_threshold=thr, noise_stddev=noise, variant=variant, posterior_mean_threashold=post,
                        scale_features=scale, sampling_propagation=sampling)
    pcfg = PlannerConfig(obs_dim=O, act_dim=A, ensemble_size=E, particles=P, n_samples=N, horizon=H, n_elite=k, iterations=I,
                         scorer=ScorerConfig(goal_slice=sp.goal_slice, observe_goal_lidar=sp.observe_goal_lidar,
                                             lidar_max_dist=sp.lidar_max_dist, goal_size=sp.goal_size,
                                             reward_distance=sp.reward_distance, reward_goal=sp.reward_goal,
                                             reward_clip=sp.reward_clip, constrain_indicator=sp.constrain_indicator,
                                             cost_kinds=list(sp.cost_kinds)),
                         act_low=pb['low'], act_high=pb['high'], n_layers=len(pb['weights'][0]['W']), smoothing=smoothing,
                         stddev_threshold=thr, noise_stddev=noise, variant=variant, posterior_mean_threashold=post,
                         sampling_propagation=sampling, scale_features=scale, world_size=world_size, rank=rank,
                         chunks_per_tile=chunks_per_tile, use_graph=use_graph)
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
