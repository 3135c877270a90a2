"""The automatic tile plan (cem_capi.hip: auto_chunks / segments_for, priced with the generated constants of csrc/cem_tile_costs.inc)
against every forced tile size on the hardware it runs on: the library's own choice must be within 3 % of the best forced one — at
the BASELINE configs and at populations nobody tuned for.  The constants are measurements of one 256-CU MI355X; this test is what
says whether they still describe the kernels (after a kernel change: scripts/sweep_chunk_costs.py --emit-table, rebuild, re-run)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# name: (obs, act, K, N, H)
CASES = {
    'B1': (60, 2, 5, 500, 25), 'B2': (60, 2, 5, 2000, 30), 'B3': (60, 2, 16, 8192, 30), 'B4': (100, 12, 8, 4096, 50), 'B5_rank': (60, 2, 5, 8192, 30),
    'off_grid_1200': (60, 2, 5, 1200, 30), 'off_grid_3100': (60, 2, 5, 3100, 20), 'off_grid_wide_1500': (100, 12, 8, 1500, 25),
    # the two planner shapes the reference ships (config/policies.yaml:2-20, models.yaml:3): 15 members; 45 / 5 particles
    'shipped_safe_cem_mpc': (60, 2, 15, 500, 8, 45), 'shipped_cem_mpc': (60, 2, 15, 150, 8, 5),
}


def _planner(pb, obs, act, K, N, H, rc, P=None):
    from ethz_safe_learning_amd import CemPlanner, PlannerConfig
    cfg = PlannerConfig(obs_dim=obs, act_dim=act, ensemble_size=K, particles=P or K, n_samples=N, horizon=H, n_elite=max(N // 10, 1), iterations=2,
                        scorer=pb['scorer'], act_low=pb['low'], act_high=pb['high'], noise_stddev=1e-3, chunks_per_tile=rc)
    pl = CemPlanner(cfg); pl.set_weights(pb['weights']); pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
    return pl


def _measure(planners, pb, rounds=5):
    """Best rollout launch (ms, HIP events on the planner's stream) of each planner, the planners taking turns round by round so that
    clock ramps and drifts hit all of them alike; the minimum over launches is what the hardware can do for that plan."""
    for pl in planners.values():
        pl.set_timing(False)
        for i in range(8):                               # (small plans: the first ~10 run at ramping clocks)
            pl.plan(pb['state'], seed=1, call=i)
        pl.set_timing(True)
    best = {k: float('inf') for k in planners}
    keys = list(planners)
    for r in range(rounds):
        for k in keys[r % len(keys):] + keys[:r % len(keys)]:     # the order rotates: whoever follows a slow, mostly idle plan is measured at lower clocks
            pl = planners[k]
            for i in range(2):
                pl.plan(pb['state'], seed=2, call=10 * r + i)
                tm = pl.last_timing()
                best[k] = min(best[k], tm['rollout_ms'] / tm['rollout_launches'])
    return best


@pytest.mark.parametrize('name', list(CASES))
def test_automatic_tile_plan_is_within_3_percent_of_the_best_forced_one(name):
    from ethz_safe_learning_amd import synthetic
    obs, act, K, N, H = CASES[name][:5]
    P = CASES[name][5] if len(CASES[name]) > 5 else None
    pb = synthetic.problem(obs, act, K)
    planners = {rc: _planner(pb, obs, act, K, N, H, rc, P) for rc in (0, 1, 2, 3, 4)}
    auto_rc, auto_segs = planners[0].tiles()[0], planners[0].segments()[0]
    # Two handles of the SAME plan differ by up to ~2.5 % (their workspaces sit at different addresses), repeatably — so the choice is
    # judged like for like, on the forced handles: the size the library picks by itself against the best size, both as forced plans.
    # A timing test on a shared pool gets a second measurement before it fails.
    for attempt in range(2):
        ms = _measure(planners, pb)
        forced = {rc: ms[rc] for rc in (1, 2, 3, 4)}
        best_rc = min(forced, key=forced.get)
        print('%s: automatic = %d chunks, %d segments: %.4f ms; forced %s' % (name, auto_rc, auto_segs, ms[0], {k: round(v, 4) for k, v in forced.items()}))
        if forced[auto_rc] <= 1.03 * forced[best_rc]:
            break
    for pl in planners.values():
        pl.close()
    if abs(ms[0] / forced[auto_rc] - 1.0) >= 0.04:
        print('note: the automatic handle and its forced twin differ by %.1f %% on this box' % (100 * abs(ms[0] / forced[auto_rc] - 1.0)))
    assert forced[auto_rc] <= 1.03 * forced[best_rc], (name, auto_rc, best_rc, forced)
