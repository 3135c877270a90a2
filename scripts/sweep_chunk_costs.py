#!/usr/bin/env python3
"""Calibration of the tile-plan cost model (cem_capi.hip: cu_cost / tile_plan_cost, constants in csrc/cem_tile_costs.inc): rollout time
of launches with exactly m tiles of rc chunks per CU (m = 1 .. 6), for both kernel families (obs+act <= 64: NFW 1; > 64: NFW 2).
Rows = CUs x m tiles x 16 rc; H = 30.  Prints one JSON line per point: ms per launch and ms per (chunk, CU) = the cost of one 16-row
chunk for the whole horizon when m workgroups share (or queue for) a CU.

  python scripts/sweep_chunk_costs.py [--cus 256] [--emit-table gpurun_out/cem_tile_costs.inc]

--emit-table regenerates the constants from THIS sweep by fixed rules (no hand fitting), so the table can be rebuilt for another
device or after a kernel change and `git diff` shows what moved:
  kChunkStart[nfw][rc][k-1]  = ms per chunk at m = k tiles per CU, k = 1 .. 3, for k up to the kernel's residency R (asked from the runtime:
                               cem_rollout_residency); entries beyond R repeat the R-th (more tiles than slots queue: they are priced by kChunkNext)
  kChunkNext[nfw][rc]        = ms per chunk at m = 6 tiles per CU (the dispatcher refilling slots as tiles retire)
  kPartialFill               = how much earlier co-resident tiles finish when only some CUs carry them: from one-chunk tiles at 1.5 per CU,
                               (1 - T / (2 kChunkStart[0][0][1])) / (1 - fill), fill = mean tiles per CU / tiles on the busiest CU
  kFloatFactor               = B2's pinned + floating-segment launch / (mean tiles per CU x kChunkStart[0][0][2]): what the floating form
                               delivers of the all-resident three-per-CU rate at 2.44 tiles per CU
Copy the file over ethz_safe_learning_amd/csrc/cem_tile_costs.inc and rebuild."""
import ctypes as C
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ethz_safe_learning_amd import CemPlanner, PlannerConfig, synthetic, _capi

CUS = int(sys.argv[sys.argv.index('--cus') + 1]) if '--cus' in sys.argv else 256
EMIT = sys.argv[sys.argv.index('--emit-table') + 1] if '--emit-table' in sys.argv else None
P = 4


def rollout_ms(cfg, pb, plans=4):
    pl = CemPlanner(cfg); pl.set_weights(pb['weights']); pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
    for i in range(3):
        pl.plan(pb['state'], seed=1, call=i)
    pl.set_timing(True)
    ms, ln = 0.0, 0
    for i in range(plans):
        pl.plan(pb['state'], seed=2, call=i); tm = pl.last_timing(); ms += tm['rollout_ms']; ln += tm['rollout_launches']
    pl.set_timing(False)
    tiles, segs = len(pl.tiles()[1]), pl.segments()[0]
    pl.close(); del pl
    return ms / ln, tiles, segs


points = {}
for nfw, O, A in ((1, 60, 2), (2, 100, 12)):
    pb = synthetic.problem(O, A, P)
    for rc in (1, 2, 3, 4):
        for m in (1, 2, 3, 4, 5, 6):
            rows = CUS * m * 16 * rc
            N = rows // P
            cfg = PlannerConfig(obs_dim=O, act_dim=A, ensemble_size=P, particles=P, n_samples=N, horizon=30, n_elite=max(N // 10, 1), iterations=2,
                                scorer=pb['scorer'], act_low=pb['low'], act_high=pb['high'], noise_stddev=1e-3, use_graph=False, chunks_per_tile=rc,
                                rollout_segments=1)
            ms, tiles, _ = rollout_ms(cfg, pb)
            assert tiles == CUS * m, (tiles, CUS * m)
            r = dict(nfw=nfw, rc=rc, tiles_per_cu=m, rollout_ms=ms, ms_per_chunk=ms / (m * rc),
                     frac=synthetic.flops_per_row_step(O, A) * rows * 30 / (ms * 1e-3) / 157.3e12)
            points[(nfw, rc, m)] = r
            print(json.dumps(r), flush=True)

if EMIT:
    lib = _capi.load()
    start = [[[0.0] * 3 for _ in range(4)] for _ in range(2)]
    nxt = [[0.0] * 4 for _ in range(2)]
    resid = [[0] * 4 for _ in range(2)]
    for nfw in (1, 2):
        for rc in (1, 2, 3, 4):
            tab, run = (C.c_int32 * 2)(), (C.c_int32 * 2)()
            _capi.check(lib.cem_rollout_residency(rc, nfw, tab, run), 'cem_rollout_residency')
            R = max(1, min(int(run[0]), 3))
            resid[nfw - 1][rc - 1] = int(run[0])
            for k in (1, 2, 3):
                start[nfw - 1][rc - 1][k - 1] = points[(nfw, rc, min(k, R))]['ms_per_chunk']
            nxt[nfw - 1][rc - 1] = points[(nfw, rc, 6)]['ms_per_chunk']
    # B2 through the automatic plan (pinned tiles + floating horizon segments)
    pb = synthetic.problem(60, 2, 5)
    cfg = PlannerConfig(obs_dim=60, act_dim=2, ensemble_size=5, particles=5, n_samples=2000, horizon=30, n_elite=200, iterations=5, scorer=pb['scorer'],
                        act_low=pb['low'], act_high=pb['high'], noise_stddev=1e-3, use_graph=False)
    b2_ms, b2_tiles, b2_segs = rollout_ms(cfg, pb, plans=8)
    ff = b2_ms / ((b2_tiles / CUS) * start[0][0][2]) if b2_segs > 1 else None
    print(json.dumps(dict(b2_rollout_ms=b2_ms, b2_tiles=b2_tiles, b2_segments=b2_segs, float_factor=ff)), flush=True)

    # one-chunk tiles at 1.5 per CU: half the CUs carry two — how much earlier than 2 x kChunkStart[0][0][1] do they finish?
    Nh = CUS * 3 // 2 * 16 // P
    pbh = synthetic.problem(60, 2, P)
    cfg = PlannerConfig(obs_dim=60, act_dim=2, ensemble_size=P, particles=P, n_samples=Nh, horizon=30, n_elite=max(Nh // 10, 1), iterations=2, scorer=pbh['scorer'],
                        act_low=pbh['low'], act_high=pbh['high'], noise_stddev=1e-3, use_graph=False, chunks_per_tile=1, rollout_segments=1)
    half_ms, half_tiles, _ = rollout_ms(cfg, pbh, plans=8)
    partial = max(0.0, min(0.6, (1.0 - half_ms / (2.0 * start[0][0][1])) / (1.0 - (half_tiles / CUS) / 2.0)))
    print(json.dumps(dict(half_filled_rollout_ms=half_ms, tiles=half_tiles, partial_fill=partial)), flush=True)

    def arr(x):
        return '{' + ', '.join(arr(v) if isinstance(v, list) else '%.4f' % v for v in x) + '}'
    with open(EMIT, 'w') as fh:
        fh.write('// GENERATED by scripts/sweep_chunk_costs.py --emit-table (%s, %s, %d CUs, library sources %s).\n'
                 % (time.strftime('%Y-%m-%d'), torch.cuda.get_device_name(0), CUS, __import__('bench').source_sha16()))
        fh.write('// ms per 16-row chunk for the whole horizon at H = 30; rules in the script header.  Residency the sweep ran at (plain kernel): %s\n' % arr([[float(v) for v in r] for r in resid]))
        fh.write('static const double kChunkStart[2][4][3] = %s;\n' % arr(start))
        fh.write('static const double kChunkNext[2][4] = %s;\n' % arr(nxt))
        fh.write('static const double kFloatFactor = %.4f;   // B2: %.4f ms with floating segments vs %.3f tiles per CU x kChunkStart[0][0][2]\n'
                 % (ff if ff else 0.945, b2_ms, b2_tiles / CUS))
        fh.write('static const double kPartialFill = %.4f;   // %d one-chunk tiles (%.2f per CU): %.4f ms vs 2 x kChunkStart[0][0][1]\n' % (partial, half_tiles, half_tiles / CUS, half_ms))
    print('wrote', EMIT)
