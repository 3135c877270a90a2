#!/usr/bin/env python3
"""Soak run: tens of thousands of plans through each kernel family on one handle each (B2 cem / the floating-segment kernel, B2 safe,
a generic-kernel shape, a handle that is re-staged with new weights every 500 plans as the agent does after `fit`), checking that
nothing drifts: a fixed (seed, call) is re-planned every 1000 plans and must return the same bits, free device memory must not
shrink, no plan may report a device fault.  usage: python scripts/soak.py [seconds per leg, default 40]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests import helpers as hp

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0


def leg(name, pb, cfg, restage_every=0, inject_at=0):
    pl = hp.make_planner(pb, cfg)
    ref = pl.plan(pb['state'], seed=11, call=7)
    free0 = torch.cuda.mem_get_info()[0]
    rng = np.random.default_rng(0)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget:
        st = pb['state'] + 0.01 * rng.standard_normal(pb['state'].shape).astype(np.float32)
        if inject_at and n == inject_at:                          # one fused select of this plan sees a grid barrier expire: recovered in stream,
            mode0 = pl.select_mode(); pl.inject_fault(1)          # the handle then plans on the multi-launch select (same bits as the fused one)
        a, s, it = pl.plan(st, seed=3, call=n)
        if inject_at and n == inject_at:
            assert mode0 == 3 and pl.select_mode() == 2, (mode0, pl.select_mode())
        assert np.all(np.isfinite(a)) and np.isfinite(s), (name, n)
        n += 1
        if restage_every and n % restage_every == 0:
            pl.set_weights(pb['weights'])                       # same values: the probe below must still match
            pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
        if n % 1000 == 0:
            a, s, it = pl.plan(pb['state'], seed=11, call=7)
            assert np.array_equal(a, ref[0]) and s == ref[1] and it == ref[2], (name, n, 'drift')
    dt = time.perf_counter() - t0
    a, s, it = pl.plan(pb['state'], seed=11, call=7)
    assert np.array_equal(a, ref[0]) and s == ref[1], (name, 'drift at end')
    free1 = torch.cuda.mem_get_info()[0]
    pl.close()
    print('%-34s %6d plans in %5.1f s (%.3f ms per plan), device memory delta %+d KiB' % (name, n, dt, 1e3 * dt / n, (free0 - free1) // 1024), flush=True)
    assert free0 - free1 < (8 << 20), 'device memory shrank by more than 8 MiB during the leg'


pb = hp.make_problem(60, 2, 5, 4, seed=1)
_, cfg = hp.configs(pb, N=2000, H=30, P=5, E=5, k=200, I=5, use_graph=True)
leg('B2 cem (floating segments, graph)', pb, cfg)
_, cfg = hp.configs(pb, N=2000, H=30, P=5, E=5, k=200, I=5, variant='safe', post=0.3, use_graph=True)
leg('B2 safe', pb, cfg)
_, cfg = hp.configs(pb, N=2000, H=30, P=5, E=5, k=200, I=5, use_graph=True)
leg('B2 cem, weights re-staged / 500', pb, cfg, restage_every=500)
_, cfg = hp.configs(pb, N=2000, H=30, P=5, E=5, k=200, I=5, use_graph=True, precision='bf16x3')
leg('B2 cem, split-product rollout', pb, cfg, restage_every=700)
pbw = hp.make_problem(60, 2, 5, 4, seed=1, units=192, activation='tanh')
_, cfg = hp.configs(pbw, N=1000, H=20, P=5, E=5, k=100, I=4, use_graph=True)
leg('generic kernel: 192 units, tanh', pbw, cfg)
_, cfg = hp.configs(pb, N=24576, H=10, P=5, E=5, k=2457, I=3, use_graph=True)
leg('fused multi-workgroup select', pb, cfg)
leg('fused select, one expiry injected', pb, cfg, inject_at=100)
pbs = hp.make_problem(60, 2, 15, 4, seed=2)
_, cfg = hp.configs(pbs, N=500, H=8, P=45, E=15, k=20, I=9, variant='safe', post=0.15, noise=0.01, use_graph=True)
leg('shipped safe_cem_mpc shape', pbs, cfg, restage_every=167)
print('soak ok')
