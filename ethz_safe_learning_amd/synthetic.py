"""Synthetic workloads of BASELINE.md: random-init ensemble weights (Keras Glorot-uniform,
mlp_ensemble.py:13,28-29), a PointGoal1-shaped observation layout (sorted keys, safety_gym.py:17) and scorer constants.
There are no datasets or checkpoints here (no network); benchmarks say "synthetic"."""
import numpy as np

from .planner import ScorerConfig


def layout(obs_dim):
    if obs_dim == 60:
        return dict(goal_lidar=(3, 19), hazards_lidar=(22, 38), vases_lidar=(41, 57))
    if obs_dim >= 32:
        return dict(goal_lidar=(0, 16), hazards_lidar=(16, 32))
    k = max(1, obs_dim // 3)
    return dict(goal_lidar=(0, k), hazards_lidar=(k, 2 * k))


def problem(obs_dim=60, act_dim=2, ensemble_size=5, units=128, n_layers=4, seed=1234, head_scale=0.05, var_bias=-8.0,
            state_seed=7):
    D = obs_dim + act_dim
    weights = []
    for m in range(ensemble_size):
        rng = np.random.default_rng(seed + m)

        def glorot(fi, fo):
            lim = np.sqrt(6.0 / (fi + fo))
            return rng.uniform(-lim, lim, size=(fi, fo)).astype(np.float32)
        Ws, bs, fi = [], [], D
        for _ in range(n_layers):
            Ws.append(glorot(fi, units)); bs.append(np.zeros((units,), np.float32)); fi = units
        weights.append(dict(W=Ws, b=bs, W_mu=(glorot(units, obs_dim) * head_scale).astype(np.float32),
                            b_mu=np.zeros((obs_dim,), np.float32),
                            W_var=(glorot(units, obs_dim) * head_scale).astype(np.float32),
                            b_var=np.full((obs_dim,), var_bias, np.float32)))
    lay = layout(obs_dim)
    lidar = np.zeros((obs_dim,), bool)
    for lo, hi in lay.values():
        lidar[lo:hi] = True
    inputs_min = np.concatenate([np.where(lidar, 0.0, -3.0), -np.ones(act_dim)]).astype(np.float32)
    inputs_max = np.concatenate([np.where(lidar, 1.0, 3.0), np.ones(act_dim)]).astype(np.float32)
    rng = np.random.default_rng(state_seed)
    state = np.where(lidar, rng.uniform(0.2, 0.9, obs_dim), rng.normal(0, 0.3, obs_dim)).astype(np.float32)
    scorer = ScorerConfig(goal_slice=lay['goal_lidar'], cost_kinds=[(lay['hazards_lidar'][0], lay['hazards_lidar'][1], 0.2)])
    return dict(weights=weights, inputs_min=inputs_min, inputs_max=inputs_max, state=state,
                low=-np.ones(act_dim, np.float32), high=np.ones(act_dim, np.float32), scorer=scorer)


def flops_per_row_step(obs_dim, act_dim, units=128, n_layers=4):
    """2*[(O+A)*U + (L-1)*U^2 + 2*U*O]  (SURVEY.md section 8d)."""
    return 2 * ((obs_dim + act_dim) * units + (n_layers - 1) * units * units + 2 * units * obs_dim)


def rehearsal_score_frames(planner, state, world, iterations, seed=0, call=0):
    """Stand-ins for the OTHER ranks' score shards when one process plays one rank of `world` on a single GPU (bench.py's B5 rehearsal,
    scripts/time_b5_rank.py): I frames of N floats, frame i = this rank's own scores of iteration i (from one stepwise plan in which
    the foreign slots hold the previous frame), tiled over the world and jittered by 1 % so that no two candidates tie.  The select's
    cost depends on how the scores are distributed (a bucket that holds the k-th key and hundreds of others takes refinement passes);
    in a real run every rank's candidates come from the same mean / stddev, so the foreign slots have to look like the rank's own —
    uncorrelated filler (standard normals next to objective sums) triples the select time at N = 65536 and is not what a node would see."""
    import torch
    N, nloc = planner.cfg.n_samples, planner.n_local
    g = torch.Generator(device='cpu'); g.manual_seed(1234 + seed)
    jitter = (1.0 + 0.01 * torch.randn(world, nloc, generator=g)).to(planner.scores_global().device)
    frames = torch.empty(iterations, N, dtype=torch.float32, device=jitter.device)
    planner.plan_begin(state, seed=seed, call=call)
    for it in range(iterations):
        planner.plan_rollout(it)
        with torch.cuda.stream(planner.stream):
            own = planner.scores_local(sync=False).clone()
            frames[it] = (own.unsqueeze(0) * jitter).reshape(N)
            planner.scores_global(sync=False).copy_(frames[it])
            planner.scores_global(sync=False)[planner.cfg.rank * nloc:(planner.cfg.rank + 1) * nloc].copy_(own)
        planner.plan_select(it)
    planner.plan_end()
    torch.cuda.synchronize()
    return frames
