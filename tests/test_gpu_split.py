"""The split-product rollout (precision 'bf16x3': csrc/cem_rollout_split.h) against the SAME oracle at the SAME tolerances as the fp32
kernels, plus the invariances that must hold bit for bit within it (rank shards, tile sizes)."""
import numpy as np
import pytest

from oracle import cem_oracle as o
from tests import helpers as hp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('variant', ['cem', 'safe'])
@pytest.mark.parametrize('dims', [(60, 2, 128), (100, 12, 128), (23, 3, 48)], ids=['obs60', 'obs100', 'obs23_units48'])
def test_whole_plan_matches_oracle(variant, dims):
    O, A, U = dims
    pb = hp.make_problem(O, A, 5, 4, seed=7 + O, units=U)
    N, H, P, E, k, I = 128, 8, 5, 5, 12, 3
    ocfg, pcfg = hp.configs(pb, N=N, H=H, P=P, E=E, k=k, I=I, variant=variant, noise=0.01, post=0.3, precision='bf16x3')
    pl = hp.make_planner(pb, pcfg)
    ea, em, eo = hp.noise(I, N, H, A, P, O, seed=3)
    a, s, it = pl.plan(pb['state'], eps_act=ea, eps_model=em, eps_out=eo)
    ra, rs, rit = o.do_generate_action(pb['state'], pb['weights'], pb['inputs_min'], pb['inputs_max'], pb['low'], pb['high'],
                                       ea, em, eo, ocfg, pb['scorer'])
    assert it == rit and abs(s - rs) <= 2e-5, (s, rs)
    np.testing.assert_allclose(a, ra, rtol=1e-5, atol=1e-6)


ATOL = 5e-6                                             # the fp32 kernels' bound (tests/test_gpu_parity.py)


def _iteration_scores(pl, pb, ea, em):
    import torch
    pl.plan_begin(pb['state'], eps_act=ea, eps_model=em)
    pl.plan_rollout(0)
    torch.cuda.synchronize()
    sc = pl.scores_local().cpu().numpy().copy()
    pl.plan_end()
    return sc


@pytest.mark.parametrize('variant', ['cem', 'safe'])
@pytest.mark.parametrize('rc', [1, 2, 3, 4])
@pytest.mark.parametrize('dims', [(60, 2), (100, 12)], ids=['obs60', 'obs100'])
def test_scores_match_oracle_and_fp32_kernel(variant, rc, dims):
    """Per-candidate scores of one iteration on dumped noise: split vs the fp64 oracle at the fp32 kernels' bound (near-threshold
    candidates under the admissible outcomes), and split vs the fp32 kernel."""
    O, A = dims
    pb = hp.make_problem(O, A, 5, 4, seed=21 + O)
    N, H, P, E, k, I = 256, 12, 5, 5, 25, 1
    ea, em, eo = hp.noise(I, N, H, A, P, O, seed=5)
    sc = {}
    for prec in ('fp32', 'bf16x3'):
        ocfg, pcfg = hp.configs(pb, N=N, H=H, P=P, E=E, k=k, I=I, variant=variant, post=0.3, precision=prec, chunks_per_tile=rc)
        pl = hp.make_planner(pb, pcfg)
        sc[prec] = _iteration_scores(pl, pb, ea, em)
        pl.close()
    lb, ub, mu0, sg0 = o.sampling_params(pb['low'], pb['high'])
    ref_actions = o.sample_actions(np.broadcast_to(mu0, (H, A)), np.broadcast_to(sg0, (H, A)), lb, ub, ea[0])
    ref64, traj64 = o.candidate_scores(pb['state'].astype(np.float64), ref_actions.astype(np.float64), o.cast_weights(pb['weights'], np.float64),
                                       pb['inputs_min'], pb['inputs_max'], em[0], ocfg, pb['scorer'], return_traj=True)
    err, n_near, n_flip = hp.assert_scores_match_oracle(sc['bf16x3'], traj64, P, N, pb['scorer'], variant, 0.3, ATOL, 'split rc %d' % rc)
    err32, _, _ = hp.assert_scores_match_oracle(sc['fp32'], traj64, P, N, pb['scorer'], variant, 0.3, ATOL, 'fp32 rc %d' % rc)
    d = np.abs(sc['bf16x3'] - sc['fp32'])
    print('obs %d %s rc %d: max |score - oracle| split %.2e, fp32 kernel %.2e; split vs fp32 kernel: median %.1e max %.1e; %d near a threshold, %d flipped'
          % (O, variant, rc, err, err32, np.median(d), d.max(), n_near, n_flip))


def test_rank_shards_and_tile_sizes_are_bit_identical_within_the_split_kernel():
    import torch
    pb = hp.make_problem(60, 2, 5, 4, seed=33)
    N, H, P, E, k, I = 512, 10, 5, 5, 51, 2
    ref = None
    for rc in (1, 2, 3, 4, 0):
        _, pcfg = hp.configs(pb, N=N, H=H, P=P, E=E, k=k, I=I, precision='bf16x3', chunks_per_tile=rc)
        pl = hp.make_planner(pb, pcfg)
        a, s, it = pl.plan(pb['state'], seed=9, call=4)
        sc = pl.scores_global().cpu().numpy().copy()
        pl.close()
        if ref is None:
            ref = (a, s, sc)
        else:
            np.testing.assert_array_equal(sc, ref[2]); np.testing.assert_array_equal(a, ref[0]); assert s == ref[1]
    # two half-shards, exchanged by hand, reproduce the single-rank scores
    halves = []
    for r in range(2):
        _, pcfg = hp.configs(pb, N=N, H=H, P=P, E=E, k=k, I=I, precision='bf16x3', world_size=2, rank=r)
        pl = hp.make_planner(pb, pcfg)
        pl.plan_begin(pb['state'], seed=9, call=4)
        pl.plan_rollout(0)
        torch.cuda.synchronize()
        halves.append(pl.scores_local().cpu().numpy().copy())
        pl.plan_end()
        pl.close()
    _, pcfg = hp.configs(pb, N=N, H=H, P=P, E=E, k=k, I=1, precision='bf16x3')
    pl = hp.make_planner(pb, pcfg)
    pl.plan(pb['state'], seed=9, call=4)
    np.testing.assert_array_equal(np.concatenate(halves), pl.scores_global().cpu().numpy())
    pl.close()


def test_unfold_sequences_on_a_split_handle_matches_the_fp32_kernels_and_the_oracle():
    """TransitionModel.unfold_sequences (transition_model.py:64-77) through the split kernel's explicit-tensor instantiation."""
    import torch
    pb = hp.make_problem(60, 2, 5, 4, seed=1)
    rng = np.random.default_rng(4)
    B, H = 40, 6
    s0 = rng.normal(0, 0.3, (B, 60)).astype(np.float32)
    acts = rng.uniform(-1, 1, (B, H, 2)).astype(np.float32)
    eps = rng.standard_normal((H, B, 60)).astype(np.float32)
    out = {}
    for prec in ('fp32', 'bf16x3'):
        _, pcfg = hp.configs(pb, N=64, H=4, P=5, E=5, k=6, I=1, precision=prec)
        pl = hp.make_planner(pb, pcfg)
        traj, mu, sd = pl.unfold_sequences(s0, acts, eps_model=eps, return_moments=True)
        torch.cuda.synchronize()
        out[prec] = (traj.cpu().numpy(), mu.cpu().numpy(), sd.cpu().numpy())
        pl.close()
    members = np.arange(B) // (B // 5)                       # mlp_ensemble.py:123-126: contiguous chunks of B / E rows
    ref = o.unfold_sequences(s0.astype(np.float64), acts.astype(np.float64), o.cast_weights(pb['weights'], np.float64), members,
                             pb['inputs_min'], pb['inputs_max'], eps)
    for prec in out:
        np.testing.assert_allclose(out[prec][0], ref, rtol=2e-5, atol=2e-5, err_msg=prec)
    for a, b in zip(out['fp32'], out['bf16x3']):
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-5)


def test_b2_full_size_on_the_split_kernel():
    """BASELINE config B2 at full size on the split kernel's own Philox draws: determinism, two half-shards = the single rank bit for
    bit, and 64 RANDOM candidates (first, last, the last member's tiles, the rest uniform) against the fp64 oracle on the dumped
    noise — both objectives."""
    from tests.test_gpu_parity import _oracle_on_candidates, _random_candidates, FULL_SIZE_ATOL
    pb = hp.make_problem(seed=1234, bias_noise=0.0)
    N, H, P, E, k, I = 2000, 30, 5, 5, 200, 2

    def scores_of(planner):
        planner.plan_begin(pb['state'], seed=3, call=1)
        planner.plan_rollout(0)
        planner.plan_end()
        return planner.scores_local().cpu().numpy().copy()
    for variant in ('cem', 'safe'):
        ocfg, pcfg = hp.configs(pb, N=N, H=H, P=P, E=E, k=k, I=I, variant=variant, post=0.3, precision='bf16x3')
        pl = hp.make_planner(pb, pcfg)
        s_a = scores_of(pl)
        np.testing.assert_array_equal(s_a, scores_of(pl))
        if variant == 'cem':
            halves = []
            for r in range(2):
                _, c2 = hp.configs(pb, N=N, H=H, P=P, E=E, k=k, I=I, precision='bf16x3', world_size=2, rank=r)
                halves.append(scores_of(hp.make_planner(pb, c2)))
            np.testing.assert_array_equal(np.concatenate(halves), s_a)
        cand, _, _ = _random_candidates(pl, N, 64, seed=17)
        a0, ref, traj = _oracle_on_candidates(pl, pb, ocfg, cand, 3, 1, P, N, H, 2, E)
        err, n_near, n_flip = hp.assert_scores_match_oracle(s_a[cand], traj, P, len(cand), pb['scorer'], variant, 0.3, FULL_SIZE_ATOL, 'B2 split ' + variant)
        print('B2 split %s: max|gpu-f64| = %.3g over %d random candidates (%d near a threshold, %d flipped)' % (variant, err, len(cand), n_near, n_flip))
        pl.close()


def test_b1_reference_scale_plan_on_the_split_kernel():
    """BASELINE config B1 (N = 500, H = 25, 5 iterations), the whole plan against the oracle on identical noise tensors."""
    pb = hp.make_problem(60, 2, 5, 4, seed=1234, bias_noise=0.0)
    N, H, P, E, k, I = 500, 25, 5, 5, 50, 5
    ocfg, pcfg = hp.configs(pb, N=N, H=H, P=P, E=E, k=k, I=I, noise=1e-3, precision='bf16x3')
    pl = hp.make_planner(pb, pcfg)
    ea, em, eo = hp.noise(I, N, H, 2, P, 60, seed=77)
    ra, rs, rit = o.do_generate_action(pb['state'], pb['weights'], pb['inputs_min'], pb['inputs_max'], pb['low'], pb['high'],
                                       ea, em, eo, ocfg, pb['scorer'])
    a, s, it = pl.plan(pb['state'], eps_act=ea, eps_model=em, eps_out=eo)
    assert it == rit and abs(s - rs) <= 2e-5, (s, rs)
    np.testing.assert_allclose(a, ra, rtol=1e-5, atol=1e-6)
    pl.close()


def test_unfold_sequences_large_batch_uses_two_chunk_tiles_on_both_precisions():
    """cem_unfold_sequences picks two-chunk tiles from 8192 rows on: the split and the fp32 kernels must still agree row for row."""
    import torch
    pb = hp.make_problem(60, 2, 5, 4, seed=2)
    rng = np.random.default_rng(6)
    B, H = 8200, 3
    s0 = rng.normal(0, 0.3, (B, 60)).astype(np.float32)
    acts = rng.uniform(-1, 1, (B, H, 2)).astype(np.float32)
    eps = rng.standard_normal((H, B, 60)).astype(np.float32)
    out = {}
    for prec in ('fp32', 'bf16x3'):
        _, pcfg = hp.configs(pb, N=64, H=4, P=5, E=5, k=6, I=1, precision=prec)
        pl = hp.make_planner(pb, pcfg)
        out[prec] = pl.unfold_sequences(s0, acts, eps_model=eps).cpu().numpy()
        torch.cuda.synchronize()
        pl.close()
    np.testing.assert_allclose(out['fp32'], out['bf16x3'], rtol=1e-5, atol=1e-5)
    members = np.arange(B) // (B // 5)
    sub = rng.choice(B, 64, replace=False)
    ref = o.unfold_sequences(s0[sub].astype(np.float64), acts[sub].astype(np.float64), o.cast_weights(pb['weights'], np.float64), members[sub],
                             pb['inputs_min'], pb['inputs_max'], eps[:, sub])
    np.testing.assert_allclose(out['bf16x3'][sub], ref, rtol=2e-5, atol=2e-5)
