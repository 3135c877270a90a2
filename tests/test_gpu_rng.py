"""Known-answer tests of the planner's random number generator (cem_device.h: philox4x32_7, cem_normal4) against an independent
numpy implementation written from the published algorithm (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11:
Philox4x32 with 7 rounds, multipliers 0xD2511F53 / 0xCD9E8D57, Weyl key increments 0x9E3779B9 / 0xBB67AE85) and from the counter /
key / uniform / Box-Muller conventions documented in include/cem_mpc.h (cem_philox_words).  Nothing here includes device code.

The reference draws its noise from TensorFlow's stateful generator (cem_mpc.py:44-47,68; mlp_ensemble.py:192-193), which cannot
be reproduced; what these tests pin is that the GPU stream is the documented function of (seed, call, counter) and nothing else."""
import numpy as np
import pytest

from tests import helpers as hp

pytestmark = pytest.mark.gpu

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_7(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(x, np.uint64) & MASK for x in (c0, c1, c2, c3))
    for r in range(7):
        p0, p1 = M0 * c0, M1 * c2                       # 32 x 32 -> 64 bit products (operands < 2^32: no uint64 overflow)
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        kr0, kr1 = np.uint64((k0 + r * W0) & 0xFFFFFFFF), np.uint64((k1 + r * W1) & 0xFFFFFFFF)
        c0, c1, c2, c3 = hi1 ^ c1 ^ kr0, lo1, hi0 ^ c3 ^ kr1, lo0
    return np.stack([c0, c1, c2, c3], -1).astype(np.uint32)


def words(seed, call, stream, it, t, sub, idx):
    idx = np.asarray(idx, np.uint64)
    k0, k1 = seed & 0xFFFFFFFF, ((seed >> 32) ^ (call >> 32)) & 0xFFFFFFFF
    full = np.broadcast_to
    return philox4x32_7(idx, full(np.uint64(t | (it << 16)), idx.shape), full(np.uint64(sub | (stream << 16)), idx.shape),
                        full(np.uint64(call & 0xFFFFFFFF), idx.shape), k0, k1)


def normals(w):
    """u = fl32(fl32(word) * 2^-32 + 2^-33) (one rounding: the products and sums below are exact in float64), then Box-Muller."""
    u = (w.astype(np.float32).astype(np.float64) * 2.0 ** -32 + 2.0 ** -33).astype(np.float32).astype(np.float64)
    ra, rb = np.sqrt(-2.0 * np.log(u[..., 0])), np.sqrt(-2.0 * np.log(u[..., 2]))
    return np.stack([ra * np.cos(2 * np.pi * u[..., 1]), ra * np.sin(2 * np.pi * u[..., 1]),
                     rb * np.cos(2 * np.pi * u[..., 3]), rb * np.sin(2 * np.pi * u[..., 3])], -1)


def test_numpy_philox_matches_the_published_known_answers():
    """Random123's kat_vectors for philox4x32 are given at 10 rounds; the same round function, run for 10 rounds here, must
    reproduce them — which pins the multipliers, the Weyl constants and the word shuffle of the numpy restatement itself."""
    def philox10(c, k):
        c0, c1, c2, c3 = (np.uint64(x) for x in c)
        for r in range(10):
            p0, p1 = M0 * c0, M1 * c2
            c0, c1, c2, c3 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64((k[0] + r * W0) & 0xFFFFFFFF), p1 & MASK, \
                             (p0 >> np.uint64(32)) ^ c3 ^ np.uint64((k[1] + r * W1) & 0xFFFFFFFF), p0 & MASK
        return [int(c0), int(c1), int(c2), int(c3)]
    assert philox10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert philox10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert philox10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def _planner(N=64, H=6, I=3):
    pb = hp.make_problem(seed=3)
    _, pcfg = hp.configs(pb, N=N, H=H, P=5, E=5, k=8, I=I)
    return pb, hp.make_planner(pb, pcfg)


@pytest.mark.parametrize('seed,call', [(0x1234567, 0), (0xDEADBEEF12345678, 0x500000007)])
def test_philox_words_bit_exact(seed, call):
    pb, pl = _planner()
    seen = set()
    for stream, it, t, sub, idx0, n in [(0, 0, 0, 0, 0, 1000), (0, 2, 5, 15, 300000, 513), (1, 1, 3, 0, 0, 64), (2, 0, 0, 0, 0, 8),
                                        (0, 65535, 65535, 65535, 0xFFFFFF00, 255)]:
        got = pl.philox_words(seed, call, stream, it, t, sub, idx0, n)
        want = words(seed, call, stream, it, t, sub, (idx0 + np.arange(n)) & 0xFFFFFFFF)
        np.testing.assert_array_equal(got, want)
        seen.update(map(bytes, got))
    assert len(seen) == 1000 + 513 + 64 + 8 + 255        # distinct counters -> distinct 128-bit outputs (Philox is a bijection per key)


@pytest.mark.parametrize('seed,call', [(11, 4), (0xABCDEF0123456789, 0x100000002)])
def test_noise_tensors_are_box_muller_of_those_words(seed, call):
    """cem_fill_noise (the streams a plan consumes: eps_act[I,N,H,A], eps_model[I,H,B,O], eps_out[A]) against numpy normals of the
    numpy words: every element, for every (iteration, row / candidate, step, quad).  v_log / v_sqrt / v_sin / v_cos are 1-ulp
    class approximations: 4e-6 absolute + 4e-6 relative (observed 1.3e-6)."""
    N, H, I, P, O, A = 64, 6, 3, 5, 60, 2
    pb, pl = _planner(N, H, I)
    ea, em, eo = (x.cpu().numpy() for x in pl.fill_noise(seed, call))
    B = P * N
    # model noise: counter (row, t | it << 16, quad | 0 << 16)
    want = np.empty((I, H, B, 64), np.float64)
    for it in range(I):
        for t in range(H):
            for fq in range(15):
                want[it, t, :, 4 * fq:4 * fq + 4] = normals(words(seed, call, 0, it, t, fq, np.arange(B)))
    np.testing.assert_allclose(em, want[..., :O], rtol=4e-6, atol=4e-6)
    want_a = np.empty((I, N, H, 4), np.float64)
    for it in range(I):
        for t in range(H):
            want_a[it, :, t, :] = normals(words(seed, call, 1, it, t, 0, np.arange(N)))
    np.testing.assert_allclose(ea, want_a[..., :A], rtol=4e-6, atol=4e-6)
    np.testing.assert_allclose(eo, normals(words(seed, call, 2, 0, 0, 0, np.arange(1)))[0, :A], rtol=4e-6, atol=4e-6)
    # moments of the big stream (115 200 draws): a gross scaling error would pass element-wise only if numpy shared it
    assert abs(em.mean()) < 0.01 and abs(em.std() - 1.0) < 0.01 and 3.5 < np.abs(em).max() < 6.8


def test_counters_of_one_plan_never_collide():
    """The counter words of every draw of a B5-sized plan (N = 65 536, P = 5, H = 30, I = 5, obs 60 -> 15 quads, act 2 -> 1 quad):
    (idx, t | it << 16, sub | stream << 16) is injective because t, it, sub < 2^16 (validated by cem_planner_create) and the
    stream tag separates model / action / output noise — checked on the packed words for the extreme indices of each stream."""
    N, P, H, I = 65536, 5, 30, 5
    seen = set()
    for stream, n_idx, n_sub in ((0, P * N, 15), (1, N, 1), (2, 1, 1)):
        for idx in sorted({0, min(1, n_idx - 1), n_idx - 1}):
            for t in ((0, 1, H - 1) if stream < 2 else (0,)):
                for it in ((0, 1, I - 1) if stream < 2 else (0,)):
                    for sub in sorted({0, n_sub - 1}):
                        c = (idx, t | (it << 16), sub | (stream << 16))
                        assert c not in seen
                        seen.add(c)
    assert (H - 1) < 2 ** 16 and (I - 1) < 2 ** 16 and 14 < 2 ** 16 and P * N < 2 ** 32
