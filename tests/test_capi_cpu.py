"""CPU-side checks of the C ABI: the library loads and exports every symbol of
include/cem_mpc.h, argument validation mirrors the reference's errors, and the
host-side layout logic (weight streams, tiles) is right.  No compute calls
(no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import cem_oracle as o
from tests.mfma_emulator import TileEmulator, dims_of

from ethz_safe_learning_amd import PlannerConfig, ScorerConfig, _capi, pack_weights_host, plan_tiles
from ethz_safe_learning_amd.planner import plan_segments
from ethz_safe_learning_amd.planner import to_c_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(**kw):
    base = dict(obs_dim=60, act_dim=2, ensemble_size=5, particles=5, n_samples=2000, horizon=30, n_elite=200,
                iterations=5, scorer=ScorerConfig(goal_slice=(3, 19), cost_kinds=[(22, 38, 0.2)]),
                act_low=[-1, -1], act_high=[1, 1])
    base.update(kw)
    return PlannerConfig(**base)


def test_library_exports_every_declared_symbol(built_lib):
    hdr = open(os.path.join(ROOT, 'include', 'cem_mpc.h')).read()
    declared = set(re.findall(r'\b(cem_[a-z_]+)\s*\(', hdr))
    declared -= {'cem_status', 'cem_variant'}
    assert declared, 'no declarations parsed'
    for name in declared:
        assert hasattr(built_lib, name), 'libcem_mpc_gfx950.so does not export %s' % name
    assert declared == set(_capi.EXPORTED_SYMBOLS)
    assert built_lib.cem_abi_version() == _capi.CEM_ABI_VERSION


def test_config_struct_matches_header_size(built_lib):
    # a wrong ctypes mirror of cem_config_t would make workspace sizes nonsense or zero
    cc = to_c_config(_cfg())
    assert built_lib.cem_workspace_bytes(C.byref(cc)) > 1 << 20
    assert built_lib.cem_weight_blob_floats(C.byref(cc)) == 5 * (62 * 128 + 128 + 3 * (128 * 128 + 128) + 2 * (128 * 60 + 60))


@pytest.mark.parametrize('kw,status', [
    (dict(units=300), 2),                                # units <= 256 (<= 128: the fast kernel, zero-padded; above: the wide kernel)
    (dict(units=0), 1),
    (dict(obs_dim=120, act_dim=12, act_low=[-1] * 12, act_high=[1] * 12), 2),                 # obs+act > 128
    (dict(particles=3, n_samples=7, n_elite=2, ensemble_size=5), 3),   # tf.split would raise (mlp_ensemble.py:123)
    (dict(n_elite=3000), 1),                             # k > N
    (dict(world_size=3), 1),                             # N % world != 0
    (dict(horizon=20000, select_mode=1), 2),             # elite list + 2 x H x A floats must fit the ONE-workgroup select kernel's LDS (an explicit request for it)
    (dict(scorer=ScorerConfig(goal_slice=(3, 61), cost_kinds=[(22, 38, 0.2)])), 1),          # goal slice runs past the observation
    (dict(scorer=ScorerConfig(goal_slice=(19, 19), cost_kinds=[(22, 38, 0.2)])), 1),         # empty goal lidar: distance would be +inf
    (dict(scorer=ScorerConfig(goal_slice=(3, 19), cost_kinds=[(38, 22, 0.2)])), 1),          # reversed cost slice
    (dict(scorer=ScorerConfig(goal_slice=(3, 19), cost_kinds=[(-1, 8, 0.2)])), 1),
    (dict(scorer=ScorerConfig(goal_slice=(60, 61), observe_goal_lidar=False)), 1),           # goal_dist feature outside the observation
    (dict(n_samples=1 << 20, n_elite=16, horizon=600, particles=5), 2),                       # N*H*A would overflow the int32 index math
])
def test_validation_errors(built_lib, kw, status):
    cc = to_c_config(_cfg(**kw))
    assert built_lib.cem_workspace_bytes(C.byref(cc)) == 0
    rc, nt = C.c_int32(), C.c_int32()
    assert built_lib.cem_plan_tiles_host(C.byref(cc), C.byref(rc), C.byref(nt), None, 0) == status
    assert built_lib.cem_status_string(status)


def test_every_status_code_has_its_own_message(built_lib):
    # include/cem_mpc.h: CEM_OK .. CEM_ERR_DEVICE (9: a kernel gave up — the floating-segment queue's bounded spin)
    msgs = [built_lib.cem_status_string(i).decode() for i in range(10)]
    assert all(msgs) and len(set(msgs)) == 10
    assert 'kernel' in msgs[9] and 'RCCL' in msgs[8]


def test_goal_threshold_crosses_the_abi_rounded_once():
    # fl32(0.3 * 0.8) evaluated in double = 0.23999999..., not fl32(0.3) * 0.8 = 0.24000001 (safety_gym.py:116)
    cc = to_c_config(_cfg())
    assert np.float32(cc.scorer.goal_reached_dist) == np.float32(0.3 * 0.8)
    assert np.float32(cc.scorer.goal_reached_dist) < np.float32(np.float32(0.3) * np.float32(0.8))


def test_null_handle_calls_fail_cleanly(built_lib):
    assert built_lib.cem_compute_objective(None, None, 0, 0, None) == 1
    assert built_lib.cem_scorer_cost(None, None, 0, None) == 1
    assert built_lib.cem_planner_destroy(None) == 1
    assert built_lib.cem_plan_rollout(None, 0) == 1
    assert built_lib.cem_planner_set_weights(None, None, 0) == 1


@pytest.mark.parametrize('obs_dim,act_dim,n_layers,rows', [(60, 2, 4, 16), (6, 2, 2, 32), (100, 12, 3, 16), (64, 2, 2, 16)])
def test_weight_stream_layout_via_mfma_lane_emulation(built_lib, obs_dim, act_dim, n_layers, rows):
    """The packed A-operand streams, consumed exactly as the kernel consumes them (lane maps of
    v_mfma_f32_16x16x4_f32, accumulator registers re-used as the next layer's B operand, 4-wave feature
    split with the LDS exchange image), reproduce x@W+b / relu of the reference MLP."""
    E = 2
    pb = o.synthetic_problem(obs_dim=obs_dim, act_dim=act_dim, ensemble_size=E, units=128, n_layers=n_layers, seed=11)
    rng = np.random.default_rng(3)
    for w in pb['weights']:                                    # non-zero biases so bias placement is tested
        for b in w['b']:
            b[:] = rng.normal(0, 0.1, b.shape)
        w['b_mu'][:] = rng.normal(0, 0.1, obs_dim)
        w['b_var'][:] = rng.normal(0, 0.1, obs_dim)
    cfg = _cfg(obs_dim=obs_dim, act_dim=act_dim, ensemble_size=E, particles=2, n_samples=64, n_elite=4, n_layers=n_layers,
               scorer=ScorerConfig(goal_slice=(0, 2)), act_low=[-1] * act_dim, act_high=[1] * act_dim)
    packed = pack_weights_host(cfg, pb['weights'])
    d = dims_of(obs_dim, act_dim, n_layers)
    assert packed.size == E * d['member_stride_f4'] * 4
    x = rng.normal(0, 1, (rows, obs_dim + act_dim))
    for m in range(E):
        w = pb['weights'][m]
        emu = TileEmulator(packed[m * d['member_stride_f4'] * 4:(m + 1) * d['member_stride_f4'] * 4].astype(np.float64), d)
        hidden, mu, var = emu.forward(x, [b.astype(np.float64) for b in w['b']], w['b_mu'].astype(np.float64),
                                      w['b_var'].astype(np.float64))
        h = x
        for l in range(n_layers):
            h = np.maximum(h @ w['W'][l].astype(np.float64) + w['b'][l], 0)
            np.testing.assert_allclose(hidden[l], h, rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(mu, h @ w['W_mu'].astype(np.float64) + w['b_mu'], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(var, h @ w['W_var'].astype(np.float64) + w['b_var'], rtol=1e-10, atol=1e-10)
        # a full time step consumed exactly one lap of every wave's stream
        assert emu.pos == [0, 0, 0, 0]


def _check_tiles(cfg, rc, tiles):
    P, N, E, W, R = cfg.particles, cfg.n_samples, cfg.ensemble_size, cfg.world_size, cfg.rank
    Nloc, n_off = N // W, R * (N // W)
    chunk = P * N // E
    seen = np.zeros(P * Nloc, int)
    for row_base, cnt, member, act_base, noise_base, s0_base in tiles:
        assert 1 <= cnt <= 16 * rc and s0_base == -1
        p, nl = divmod(row_base, Nloc)
        assert nl + cnt <= Nloc                                   # a tile stays inside one particle
        assert act_base == n_off + nl                             # global candidate index
        assert noise_base == p * N + n_off + nl                   # global row id
        for r in (noise_base, noise_base + cnt - 1):
            assert r // chunk == member                           # member(r) = r // (B/E), mlp_ensemble.py:123-126
        seen[row_base:row_base + cnt] += 1
    assert np.all(seen == 1)                                      # every local row exactly once


@pytest.mark.parametrize('kw', [
    dict(),                                                       # B2
    dict(n_samples=500, horizon=25, n_elite=50),                  # B1
    dict(ensemble_size=16, particles=16, n_samples=8192, n_elite=819),   # B3
    dict(ensemble_size=15, particles=5, n_samples=150, n_elite=15),      # shipped cem_mpc: candidates of a particle hit 3 members
    dict(ensemble_size=15, particles=45, n_samples=500, n_elite=20),     # shipped safe_cem_mpc: 3 particles per member
    dict(n_samples=65536, n_elite=6554, world_size=8, rank=3),    # B5 shard
    dict(ensemble_size=3, particles=2, n_samples=9, n_elite=2, world_size=3, rank=2),  # ragged: member boundary inside a shard
    dict(chunks_per_tile=2),
])
def test_tiles_partition_rows_and_respect_member_boundaries(built_lib, kw):
    cfg = _cfg(**kw)
    rc, tiles = plan_tiles(cfg)
    assert 1 <= rc <= 4
    if cfg.chunks_per_tile:
        assert rc == cfg.chunks_per_tile
    _check_tiles(cfg, rc, tiles)


def test_tile_size_choice_follows_the_measured_cost_model(built_lib):
    """cem_capi.hip auto_chunks: max chunks per CU x per-chunk cost (cheaper when >= 2 workgroups share the CU).  Without a
    device the residency table is the static one; tests/test_gpu_parity.py checks it against the runtime's answer."""
    rc, tiles = plan_tiles(_cfg())                                                  # B2: 625 one-chunk tiles, <= 3 per CU
    assert rc == 1 and len(tiles) == 625
    rc, tiles = plan_tiles(_cfg(n_samples=500, horizon=25, n_elite=50))             # B1: one chunk per CU
    assert rc == 1 and len(tiles) <= 256
    rc, tiles = plan_tiles(_cfg(ensemble_size=16, particles=16, n_samples=8192, n_elite=819))   # B3: 8 four-chunk tiles per CU, 2 resident
    assert rc == 4 and len(tiles) == 2048
    rc, tiles = plan_tiles(_cfg(n_samples=8192, n_elite=819))                       # B5 shard: exactly 5 two-chunk tiles per CU
    assert rc == 2 and len(tiles) == 1280
    rc, tiles = plan_tiles(_cfg(ensemble_size=15, particles=45, n_samples=500, n_elite=20, horizon=8))   # shipped safe_cem_mpc: 22 500 rows
    assert rc in (2, 3)                                    # (a measured tie: 0.224 / 0.223 ms, tests/test_gpu_tileplan.py; the round-4 table picks 2)
    rc, tiles = plan_tiles(_cfg(n_samples=1200, n_elite=120))                       # 375 one-chunk tiles: half the CUs carry two (kPartialFill), not 188 two-chunk tiles
    assert rc == 1 and len(tiles) == 375


def test_tile_size_choice_moves_with_the_cu_count(built_lib):
    """The cost model is per CU, so it prices a plan for any CU count (cem_capi.hip device_facts: multiProcessorCount; CEM_ASSUME_CUS for
    the GPU-less helpers): B2's 10 000 rows are 2.4 one-chunk tiles per CU on 256 CUs, but 9.8 on 64 — there larger tiles (fewer weight
    streams per row) win, and no floating segments are planned when a CU holds more tiles than it keeps resident."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from tests.test_capi_cpu import _cfg; from ethz_safe_learning_amd.planner import plan_tiles, plan_segments; "
            "c = _cfg(); print(plan_tiles(c)[0], len(plan_tiles(c)[1]), plan_segments(c)[0])" % ROOT)
    out = {}
    for cus in (64, 128, 256, 304):
        r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, CEM_ASSUME_CUS=str(cus)), capture_output=True, text=True, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        out[cus] = tuple(int(x) for x in r.stdout.split())
    assert out[256] == (1, 625, 6) and out[304][0] == 1            # the shipped device; a few more CUs: still one-chunk tiles
    assert out[64][0] >= 2 and out[64][2] == 1, out                  # a quarter of the CUs: bigger tiles, plain launch
    assert out[64][0] >= out[128][0] >= out[256][0], out


def test_horizon_segments_are_chosen_where_tiles_do_not_divide_the_cus(built_lib):
    # B2: 625 one-chunk tiles on 256 CUs (3 on the busiest, 2.44 mean): 512 stay whole, 113 float in six 5-step segments
    assert plan_segments(_cfg()) == (6, 5)
    assert plan_tiles(_cfg())[0] == 1
    # one pinned tile per CU (375 tiles) or a remainder that nearly fills the CUs (750 tiles): measured no gain -> not used
    assert plan_segments(_cfg(n_samples=1200, n_elite=120))[0] == 1
    assert plan_segments(_cfg(n_samples=2400, n_elite=240))[0] == 1
    assert plan_segments(_cfg(n_samples=1800, n_elite=180))[0] == 6          # 563 tiles: 2.2 per CU
    # B1 (157 tiles: at most one per CU), B3 / B4 (tile counts that are multiples of 256): one workgroup per tile, as before
    assert plan_segments(_cfg(n_samples=500, horizon=25, n_elite=50))[0] == 1
    assert plan_segments(_cfg(ensemble_size=16, particles=16, n_samples=8192, n_elite=819))[0] == 1
    assert plan_segments(_cfg(obs_dim=100, act_dim=12, ensemble_size=8, particles=8, n_samples=4096, horizon=50, n_elite=409,
                              act_low=[-1] * 12, act_high=[1] * 12, scorer=ScorerConfig(goal_slice=(0, 16), cost_kinds=[(16, 32, 0.2)])))[0] == 1
    # explicit requests: off, and a count the horizon cannot fill is reduced so that no segment is empty
    assert plan_segments(_cfg(rollout_segments=1)) == (1, 30)
    assert plan_segments(_cfg(rollout_segments=4)) == (4, 8)
    assert plan_segments(_cfg(horizon=7, rollout_segments=5)) == (4, 2)
    assert plan_segments(_cfg(horizon=3, rollout_segments=8)) == (3, 1)
    # short horizons are never segmented automatically
    assert plan_segments(_cfg(horizon=8))[0] == 1


def test_xcd_order_groups_members(built_lib):
    # blocks b, b+8, ... share an XCD: each XCD should see few distinct members (weight sets) in its L2
    rc, tiles = plan_tiles(_cfg(ensemble_size=16, particles=16, n_samples=8192, n_elite=819))
    for x in range(8):
        members = set(tiles[x::8, 2].tolist())
        assert len(members) <= 3


# ---- property-based checks of the host logic (hypothesis) ---------------------------------------------------------------
from hypothesis import given, settings, strategies as st


@settings(max_examples=60, deadline=None)
@given(E=st.integers(1, 6), ppm=st.integers(1, 3), nmul=st.integers(1, 40), world=st.sampled_from([1, 2, 4]), chunks=st.integers(0, 4),
       obs=st.sampled_from([40, 60, 100]), data=st.data())
def test_tiles_partition_property(built_lib, E, ppm, nmul, world, chunks, obs, data):
    """For arbitrary (E, P, N, world, rank, tile size): tiles partition the rank's rows exactly once, never cross a member
    boundary (member(r) = r // (B/E), mlp_ensemble.py:123-126), and carry the global noise row / action indices."""
    P, N = E * ppm, nmul * world                       # P a multiple of E keeps P*N divisible by E (tf.split)
    rank = data.draw(st.integers(0, world - 1))
    cfg = _cfg(obs_dim=obs, act_dim=2, ensemble_size=E, particles=P, n_samples=N, n_elite=max(1, N // 10), world_size=world,
               rank=rank, chunks_per_tile=chunks)
    rc, tiles = plan_tiles(cfg)
    assert 1 <= rc <= 4 and (chunks == 0 or rc == chunks)
    _check_tiles(cfg, rc, tiles)


def test_makefile_rebuilds_on_any_header_change(tmp_path):
    """A header missing from the library target's prerequisites means timing (or testing) a stale binary after a
    header-only edit — that happened with cem_train.h.  `make -q` must report 'out of date' when any header is newer."""
    import glob
    import shutil
    import subprocess
    import time
    src = os.path.join(ROOT, 'ethz_safe_learning_amd', 'csrc')
    work = tmp_path / 'ethz_safe_learning_amd' / 'csrc'
    shutil.copytree(src, work)
    shutil.copytree(os.path.join(ROOT, 'include'), tmp_path / 'include')
    (tmp_path / 'ethz_safe_learning_amd' / 'lib').mkdir()
    lib = tmp_path / 'ethz_safe_learning_amd' / 'lib' / 'libcem_mpc_gfx950.so'
    headers = sorted(glob.glob(str(work / '*.h')) + glob.glob(str(tmp_path / 'include' / '*.h')))
    assert len(headers) >= 3
    for h in headers:
        now = time.time()
        for f in glob.glob(str(work / '*')) + headers:
            os.utime(f, (now - 100, now - 100))
        lib.write_bytes(b'')                                  # a "built" library, newer than every source
        os.utime(lib, (now - 50, now - 50))
        assert subprocess.run(['make', '-q', '-C', str(work)]).returncode == 0, 'up to date expected'
        os.utime(h, (now, now))                               # touch one header
        assert subprocess.run(['make', '-q', '-C', str(work)]).returncode == 1, '%s is not a prerequisite' % os.path.basename(h)
