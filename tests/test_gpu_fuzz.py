"""Randomised shapes against the oracle: the fixed cases of test_gpu_parity.py / test_gpu_training.py cover the shipped and the
BASELINE shapes; these sweep the corners nobody picked by hand (odd widths and depths, one-row tiles, obs+act on both sides of
64, members that split a particle, every tile size, forced horizon segments, both objectives, sampling / scaling off) with a
fixed seed per case, so a failure names a reproducible configuration.  Oracle = oracle/cem_oracle.py in fp64 (PARITY UNPINNED:
the oracle is this repo's restatement of the reference)."""
import os

import numpy as np
import pytest

from oracle import cem_oracle as o
from tests import helpers as hp
from tests.test_gpu_parity import _run_iteration, _score_err, ATOL

pytestmark = pytest.mark.gpu
# CEM_FUZZ_SCALE=n multiplies the number of seeds of every sweep (a longer one-off hunt; the default suite stays seconds long)
SCALE = int(os.environ.get('CEM_FUZZ_SCALE', '1'))


def _random_rollout_case(seed):
    rng = np.random.default_rng(1000 + seed)
    A = int(rng.integers(1, 7))
    O = int(rng.choice([5, 9, 17, 31, 40, 60, 62, 63, 70, 97, 110, 122]))
    O = min(O, 128 - A)
    E = int(rng.integers(1, 7))
    per = int(rng.integers(1, 4))                         # particles per member, or members per particle
    if rng.random() < 0.5:
        P, N = E * per, int(rng.integers(3, 120))          # whole particles per member
    else:
        P = int(rng.integers(1, 5))                        # members split particles: P * N must divide by E
        N = E * int(rng.integers(1, 40))
    c = dict(O=O, A=A, E=E, P=P, N=N, H=int(rng.integers(1, 11)), L=int(rng.integers(1, 6)),
             units=int(rng.choice([16, 17, 33, 64, 100, 127, 128, 129, 160, 200, 256])), variant=str(rng.choice(['cem', 'safe'])),
             rc=int(rng.integers(0, 5)), seg=int(rng.choice([0, 0, 1, 2, 3])), sampling=bool(rng.random() < 0.8),
             scale=bool(rng.random() < 0.8), post=float(rng.choice([0.15, 0.3, 0.5])))
    # (drawn last, so the cases above are the ones of earlier rounds) the split-product rollout where it applies: half of the eligible cases
    c['precision'] = 'bf16x3' if (c['units'] <= 128 and rng.random() < 0.5) else 'fp32'
    # (round 4, drawn after everything else) the action Box: per-dimension bounds, unbounded, one infinite bound (mpc_policy.py:45-57)
    c['box'], c['low'], c['high'] = hp.random_action_bounds(rng, A)
    return c


@pytest.mark.parametrize('seed', range(32 * SCALE))
def test_random_shape_rollout_scores(seed):
    c = _random_rollout_case(seed)
    O, A, E, P, N, H = c['O'], c['A'], c['E'], c['P'], c['N'], c['H']
    pb = hp.with_action_bounds(hp.make_problem(O, A, E, c['L'], seed=200 + seed, units=c['units']), c['low'], c['high'])
    ocfg, pcfg = hp.configs(pb, N=N, H=H, P=P, E=E, k=max(1, N // 10), I=1, variant=c['variant'], post=c['post'],
                            sampling=c['sampling'], scale=c['scale'], chunks_per_tile=c['rc'], rollout_segments=c['seg'],
                            precision=c['precision'])
    pl = hp.make_planner(pb, pcfg)
    ea, em, eo = hp.noise(1, N, H, A, P, O, seed=seed)
    actions, returns, scores = _run_iteration(pl, pb, ocfg, ea, em)
    lb, ub, mu0, sg0 = o.sampling_params(pb['low'], pb['high'])
    ref_actions = o.sample_actions(np.broadcast_to(mu0, (H, A)), np.broadcast_to(sg0, (H, A)), lb, ub, ea[0])
    np.testing.assert_array_equal(actions, ref_actions, err_msg=str(c))
    w64 = o.cast_weights(pb['weights'], np.float64)
    ref64, traj64 = o.candidate_scores(pb['state'].astype(np.float64), ref_actions.astype(np.float64), w64, pb['inputs_min'],
                                       pb['inputs_max'], em[0], ocfg, pb['scorer'], return_traj=True)
    hp.assert_scores_match_oracle(scores, traj64, P, N, pb['scorer'], c['variant'], c['post'], ATOL, str(c))   # near-threshold candidates included
    pl.close()


def _random_training_case(seed):
    rng = np.random.default_rng(5000 + seed)
    O = int(rng.choice([3, 6, 17, 28, 60, 61, 100, 120]))
    A = int(rng.integers(1, 8))
    return dict(E=int(rng.integers(1, 6)), D=min(O + A, 128), O=O, L=int(rng.integers(1, 8)), bt=int(rng.integers(1, 65)),
                units=int(rng.choice([8, 17, 31, 48, 64, 99, 128, 144, 201, 256])), kernel=str(rng.choice(['tile', 'gemm'])))


@pytest.mark.parametrize('seed', range(16 * SCALE))
def test_random_shape_training_steps(seed, monkeypatch):
    import torch
    from ethz_safe_learning_amd.trainer import CemTrainer
    c = _random_training_case(seed)
    if c['kernel'] == 'gemm':
        monkeypatch.setenv('CEM_TRAIN_GEMM_KERNEL', '1')
    else:
        monkeypatch.delenv('CEM_TRAIN_GEMM_KERNEL', raising=False)
    E, D, O, L, bt, units = c['E'], c['D'], c['O'], c['L'], c['bt'], c['units']
    pb = hp.make_problem(O, D - O, E, L, seed=300 + seed, bias_noise=0.05, head_scale=0.3, var_bias=-2.0, units=units)
    rng = np.random.default_rng(seed)
    n = 300
    X = rng.normal(0, 0.5, (n, D)).astype(np.float32)
    Y = (0.1 * X[:, :O] + 0.05 * rng.normal(0, 1, (n, O))).astype(np.float32)
    tr = CemTrainer(D, O, units, L, E, batch_size=64)
    tr.set_state(pb['weights'])
    w64 = o.cast_weights(pb['weights'], np.float64)
    ms64, vs64 = o.zeros_like_weights(w64), o.zeros_like_weights(w64)
    w32 = o.cast_weights(pb['weights'], np.float32)                       # how far fp32 arithmetic itself drifts from fp64 on this case
    ms32, vs32 = o.zeros_like_weights(w32), o.zeros_like_weights(w32)
    x_dev, y_dev = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
    lr = 0.00025
    for t in range(1, 4):
        perm = np.stack([rng.permutation(n) for _ in range(E)]).astype(np.int32)
        perm_dev = torch.from_numpy(perm).cuda()
        loss_dev = torch.zeros(E, device='cuda')
        off = 5 * t
        tr.step(x_dev, y_dev, perm_dev, off, bt, lr, loss_dev)
        tr.synchronize()
        idx = perm[:, off:off + bt]
        ref = o.training_step(w64, ms64, vs64, X[idx].astype(np.float64), Y[idx].astype(np.float64), lr, t)
        o.training_step(w32, ms32, vs32, X[idx], Y[idx], np.float32(lr), t)
        got = float(loss_dev.sum().item())
        assert abs(got - ref) <= 1e-5 * max(1.0, abs(ref)), (c, t, got, ref)

    def worst_of(wa, wb):
        return max(float(np.abs(ka - kb).max()) for a, b in zip(wa, wb) for ka, kb in zip(o._flat_params(a), o._flat_params(b)))
    # Adam's first steps move every weight by ~lr whatever the gradient's size, so a gradient of ~1e-12 turns rounding into the
    # SIGN of an update: where the numpy fp32 oracle itself drifts from fp64 (deep, wide nets), the same allowance applies
    # (and the GPU's summation order can flip one the numpy order does not: 1 case in 4 400 landed at 2.2e-5).  A wrong gradient
    # moves weights by ~lr per step = 7.5e-4 over these three steps, so 6e-5 still separates rounding from a bug by a decade.
    worst, drift32 = worst_of(tr.get_weights(), w64), worst_of(w32, w64)
    assert worst <= max(6e-5, 3.0 * drift32), (c, worst, drift32)
    vl = tr.validation_loss(x_dev[:77], y_dev[:77])
    ref_vl = o.validation_loss(w64, X[:77].astype(np.float64), Y[:77].astype(np.float64))
    assert abs(vl - ref_vl) <= 2e-5 * max(1.0, abs(ref_vl)), (c, vl, ref_vl)
    tr.close()


def _random_plan_case(seed):
    rng = np.random.default_rng(9000 + seed)
    A = int(rng.integers(1, 5))
    O = int(rng.choice([9, 31, 60, 70, 100]))
    E = int(rng.integers(1, 5))
    P = E * int(rng.integers(1, 3))
    N = int(rng.integers(2, 400))
    k = int(rng.choice([1, 2, max(1, N // 10), max(1, N // 2), N]))
    c = dict(O=O, A=A, E=E, P=P, N=N, k=min(k, N), H=int(rng.integers(1, 9)), I=int(rng.integers(2, 5)),
             variant=str(rng.choice(['cem', 'safe'])), smoothing=float(rng.choice([0.0, 0.1, 0.5])),
             thr=float(rng.choice([-1.0, -1.0, 0.3, 0.6])), noise=float(rng.choice([0.0, 0.05])),
             select_mode=int(rng.choice([0, 1, 2, 3])), use_graph=False, units=int(rng.choice([64, 64, 192])))
    c['box'], c['low'], c['high'] = hp.random_action_bounds(rng, A)          # (round 4, drawn last: earlier rounds' shapes unchanged)
    return c


@pytest.mark.parametrize('seed', range(24 * SCALE))
def test_random_shape_whole_plan_teacher_forced(seed):
    """Every iteration of a stepwise plan, each stage checked against the oracle on the GPU's OWN inputs (so a near-tie can never
    excuse a mismatch): sampled actions from the GPU's mu / sigma (bit-exact), the elite SET from the GPU's scores (exact, ties
    to the lower index), mu / sigma after the refit, best-so-far, the early-stop decision and the returned action."""
    import torch
    c = _random_plan_case(seed)
    O, A, E, P, N, H, I, k = c['O'], c['A'], c['E'], c['P'], c['N'], c['H'], c['I'], c['k']
    pb = hp.with_action_bounds(hp.make_problem(O, A, E, 2, seed=400 + seed, units=c['units']), c['low'], c['high'])
    ocfg, pcfg = hp.configs(pb, N=N, H=H, P=P, E=E, k=k, I=I, variant=c['variant'], post=0.3, smoothing=c['smoothing'], thr=c['thr'],
                            noise=c['noise'], select_mode=c['select_mode'])
    pl = hp.make_planner(pb, pcfg)
    ea, em, eo = hp.noise(I, N, H, A, P, O, seed=seed)
    lb, ub, mu0, sg0 = o.sampling_params(pb['low'], pb['high'])
    pl.plan_begin(pb['state'], eps_act=ea, eps_model=em)
    best, best_score = np.zeros(A, np.float32), np.float32(-np.inf)
    ms = np.stack([np.broadcast_to(mu0, (H, A)), np.broadcast_to(sg0, (H, A))]).astype(np.float32)
    ran = 0
    for it in range(I):
        pl.plan_rollout(it)
        torch.cuda.synchronize()
        actions = pl.actions().cpu().numpy().copy()
        scores = pl.scores_local().cpu().numpy().copy()
        np.testing.assert_array_equal(actions, o.sample_actions(ms[0], ms[1], lb, ub, ea[it]), err_msg=str((c, it)))
        assert np.all(np.isfinite(scores)), (c, it)
        pl.plan_select(it)
        torch.cuda.synchronize()
        ran += 1
        mu, sigma, best, best_score, ref_elite, stop = o.select_and_refit(scores, actions, ms[0], ms[1], best, best_score, ocfg)
        np.testing.assert_array_equal(np.sort(pl.elite_idx().cpu().numpy()), np.sort(ref_elite), err_msg=str((c, it)))
        got = pl.mu_sigma().cpu().numpy().copy()
        amag = max(1.0, float(np.abs(ub).max()))             # mu / sigma live on the scale of the Box (+-100 when it is unbounded)
        np.testing.assert_allclose(got[0], mu, rtol=1e-5, atol=1e-6 * amag, err_msg=str((c, it)))
        # a dimension whose samples are all the same value c (a one-point Box dimension: sigma0 = 0) has a standard deviation of pure
        # rounding noise — the mean of k copies of c is not c — of the order sqrt(k) eps |c| in either summation order (60x hunt, round 4)
        sig_atol = 1e-6 * amag + 4 * 1.2e-7 * float(np.abs(mu).max()) * np.sqrt(k)
        np.testing.assert_allclose(got[1], sigma, rtol=2e-5, atol=sig_atol, err_msg=str((c, it)))
        ms = got                                        # teacher forcing: the next iteration samples from the GPU's own refit
        # the stop rule compares mean(sigma) with the threshold: only decisive margins are asserted
        margin = abs(float(sigma.mean()) - c['thr'])
        if stop and margin > 1e-5:
            break
        if not stop and margin <= 1e-5:
            break                                       # undecidable at fp32: stop comparing here
    a, s, n_it = pl.plan_end(eps_out=eo)
    if margin > 1e-5:
        assert n_it == ran, (c, n_it, ran)
        np.testing.assert_array_equal(a, best + eo * np.float32(c['noise']), err_msg=str(c))
        assert s == best_score, (c, s, best_score)
    pl.close()


def _random_shard_case(seed):
    rng = np.random.default_rng(13000 + seed)
    W = int(rng.choice([2, 3, 4, 8]))
    E = int(rng.integers(1, 6))
    P = E * int(rng.integers(1, 3)) if rng.random() < 0.6 else int(rng.integers(1, 5))
    per = int(rng.integers(1, 60))
    N = W * per
    if (P * N) % E:
        N = W * per * E
    A = int(rng.integers(1, 4))
    O = int(rng.choice([9, 40, 60, 100]))
    c = dict(W=W, E=E, P=P, N=N, O=O, A=A, H=int(rng.integers(1, 9)), variant=str(rng.choice(['cem', 'safe'])),
             rc_full=int(rng.integers(0, 5)), rc_shard=int(rng.integers(0, 5)), seg=int(rng.choice([0, 1, 2, 3])),
             units=int(rng.choice([96, 96, 176])))
    c['precision'] = 'bf16x3' if (c['units'] <= 128 and rng.random() < 0.5) else 'fp32'     # (drawn last: earlier rounds' cases unchanged)
    c['box'], c['low'], c['high'] = hp.random_action_bounds(rng, A)                           # (round 4)
    return c


@pytest.mark.parametrize('seed', range(16 * SCALE))
def test_random_shape_shard_and_tile_invariance(seed):
    """Philox mode: the scores (and cost bytes) of a population do not depend on how it is cut — rank shards of a world of W
    (each with its own handle, tile size and segment count) concatenate to the single-rank result bit for bit: tiles, members
    and noise are keyed on GLOBAL candidate / row indices."""
    import torch
    c = _random_shard_case(seed)
    W, E, P, N, O, A, H = c['W'], c['E'], c['P'], c['N'], c['O'], c['A'], c['H']
    pb = hp.with_action_bounds(hp.make_problem(O, A, E, 3, seed=500 + seed, units=c['units']), c['low'], c['high'])

    def run(world, rank, rc, seg):
        _, pcfg = hp.configs(pb, N=N, H=H, P=P, E=E, k=max(1, N // 10), I=1, variant=c['variant'], post=0.3, world_size=world, rank=rank,
                             chunks_per_tile=rc, rollout_segments=seg, precision=c['precision'])
        pl = hp.make_planner(pb, pcfg)
        pl.plan_begin(pb['state'], seed=77, call=seed)
        pl.plan_rollout(0)
        torch.cuda.synchronize()
        out = (pl.scores_local().cpu().numpy().copy(), pl.returns().cpu().numpy().copy(),
               pl.costs().cpu().numpy().copy() if c['variant'] == 'safe' else None)
        pl.close()
        return out
    full = run(1, 0, c['rc_full'], 0)
    assert np.isfinite(full[0]).all(), c
    parts = [run(W, r, c['rc_shard'], c['seg']) for r in range(W)]
    np.testing.assert_array_equal(np.concatenate([p_[0] for p_ in parts]), full[0], err_msg=str(c))
    # per-row returns are particle-major within a rank: [P][N/W]
    Nl = N // W
    ret = np.concatenate([p_[1].reshape(P, Nl) for p_ in parts], axis=1).reshape(-1)
    np.testing.assert_array_equal(ret, full[1].reshape(-1), err_msg=str(c))
    if c['variant'] == 'safe':
        cost = np.concatenate([p_[2].reshape(H, P, Nl) for p_ in parts], axis=2).reshape(H, -1)
        np.testing.assert_array_equal(cost, full[2].reshape(H, -1), err_msg=str(c))


def _random_unfold_case(seed):
    rng = np.random.default_rng(17000 + seed)
    A = int(rng.integers(1, 9))
    O = min(int(rng.choice([3, 6, 20, 47, 60, 64, 65, 100, 120])), 128 - A)
    E = int(rng.integers(1, 7))
    return dict(O=O, A=A, E=E, L=int(rng.integers(1, 6)), B=E * int(rng.integers(1, 70)), H=int(rng.integers(1, 8)),
                units=int(rng.choice([16, 40, 64, 101, 128, 130, 224, 256])), sampling=bool(rng.random() < 0.7), scale=bool(rng.random() < 0.7),
                variant=str(rng.choice(['cem', 'safe'])))


@pytest.mark.parametrize('seed', range(16 * SCALE))
def test_random_shape_unfold_and_objective_ops(seed):
    """TransitionModel.unfold_sequences on arbitrary per-row start states (cem_unfold_sequences) against the fp64 oracle, and
    compute_objective on the trajectory it returns against the oracle's objective on the same tensor."""
    c = _random_unfold_case(seed)
    O, A, E, L, B, H = c['O'], c['A'], c['E'], c['L'], c['B'], c['H']
    pb = hp.make_problem(O, A, E, L, seed=600 + seed, units=c['units'])
    P = E
    ocfg, pcfg = hp.configs(pb, N=B // P, H=H, P=P, E=E, k=1, I=1, sampling=c['sampling'], scale=c['scale'], variant=c['variant'], post=0.5)
    pl = hp.make_planner(pb, pcfg)
    rng = np.random.default_rng(seed)
    s0 = (pb['state'][None, :] + 0.05 * rng.standard_normal((B, O))).astype(np.float32)
    acts = rng.uniform(-1, 1, (B, H, A)).astype(np.float32)
    eps = rng.standard_normal((H, B, O)).astype(np.float32)
    traj_dev = pl.unfold_sequences(s0, acts, eps_model=eps)
    traj = traj_dev.cpu().numpy()
    members = o.member_of_rows(B, E)
    w64 = o.cast_weights(pb['weights'], np.float64)
    ref64 = o.unfold_sequences(s0.astype(np.float64), acts.astype(np.float64), w64, members, pb['inputs_min'], pb['inputs_max'],
                               eps.astype(np.float64), c['scale'], c['sampling'])
    np.testing.assert_array_equal(traj[:, 0], s0, err_msg=str(c))
    assert np.abs(traj - ref64).max() <= 5e-6 * max(1.0, np.abs(ref64).max()), (c, float(np.abs(traj - ref64).max()))
    got = pl.compute_objective(traj_dev).cpu().numpy()
    t64 = traj.astype(np.float64)
    N = B // P
    ref = o.compute_objective_safe(t64, P, N, pb['scorer'], 0.5) if c['variant'] == 'safe' else o.compute_objective_cem(t64, P, N, pb['scorer'])
    ok = o.threshold_margins(t64, pb['scorer']).reshape(P, N).min(axis=0) > 1e-5
    if ok.any():
        assert _score_err(got[ok], ref[ok]) <= 1.0, (c, float(np.abs(got - ref)[ok].max()))
    pl.close()
