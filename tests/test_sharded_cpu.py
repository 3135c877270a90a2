"""The N>1 path on CPU: world_size-2 gloo processes drive ShardedCemDriver (the product's exchange logic) over an
oracle-backed backend; every rank must reproduce the single-process plan bit for bit (one all-gather of
scores per CEM iteration, everything else replicated)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cem_oracle as o
from tests import helpers as hp
from tests.oracle_backend import OracleBackend

from ethz_safe_learning_amd.sharded import ShardedCemDriver


def _problem(variant):
    pb = o.synthetic_problem(obs_dim=6, act_dim=2, ensemble_size=2, units=16, n_layers=2, seed=3)
    ocfg = o.PlanConfig(horizon=4, iterations=3, n_samples=24, n_elite=5, particles=4, ensemble_size=2, noise_stddev=0.05,
                        variant=variant, posterior_mean_threashold=0.45)
    ea, em, eo = hp.noise(3, 24, 4, 2, 4, 6, seed=21)
    return pb, ocfg, ea, em, eo


def _worker(rank, world, port, variant, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        pb, ocfg, ea, em, eo = _problem(variant)
        be = OracleBackend(pb, ocfg, world, rank)
        drv = ShardedCemDriver(be, ocfg.iterations, world_size=world)
        a, s, it = drv.plan(pb['state'], eps_act=ea, eps_model=em, eps_out=eo)
        out[rank] = (a, s, it, [t['scores'] for t in be.trace], [t['elite'] for t in be.trace])
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize('variant', ['cem', 'safe'])
def test_two_rank_plan_equals_single_process(variant):
    pb, ocfg, ea, em, eo = _problem(variant)
    trace = []
    ra, rs, rit = o.do_generate_action(pb['state'], pb['weights'], pb['inputs_min'], pb['inputs_max'], pb['low'], pb['high'],
                                       ea, em, eo, ocfg, pb['scorer'], trace=trace)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), variant, out), nprocs=2, join=True)
    assert set(out.keys()) == {0, 1}
    for rank in (0, 1):
        a, s, it, scores, elites = out[rank]
        assert it == rit
        np.testing.assert_array_equal(a, ra)
        assert s == float(rs)
        for i in range(it):
            np.testing.assert_array_equal(scores[i], trace[i]['scores'])       # gathered scores == full-batch scores
            np.testing.assert_array_equal(elites[i], trace[i]['elite'])


def test_single_rank_driver_needs_no_process_group():
    pb, ocfg, ea, em, eo = _problem('cem')
    be = OracleBackend(pb, ocfg, 1, 0)
    a, s, it = ShardedCemDriver(be, ocfg.iterations, world_size=1).plan(pb['state'], eps_act=ea, eps_model=em, eps_out=eo)
    ra, rs, rit = o.do_generate_action(pb['state'], pb['weights'], pb['inputs_min'], pb['inputs_max'], pb['low'], pb['high'],
                                       ea, em, eo, ocfg, pb['scorer'])
    np.testing.assert_array_equal(a, ra)
    assert it == rit
