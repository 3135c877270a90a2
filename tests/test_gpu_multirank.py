"""The library's multi-rank path with world_size > 1 ON HARDWARE, several ranks sharing the one GPU of a lease.

RCCL refuses two ranks on one device (scripts/try_two_ranks_one_gpu.py), so the collective — and only the collective — is
replaced by tests/fakes/libfake_rccl.so (CEM_RCCL_LIBRARY): a shared-memory all-gather built from stream operations.  Everything
else is the product path: cem_planner_comm_init, candidate shards, the ncclAllGather call inside cem_planner_plan, the captured
hipGraph with the collective in it, the replicated select, cem_plan_exchange in the stepwise form, and bench.py's multi-rank
leg.  Pass criterion: every rank returns bit for bit what a single-rank planner of the same configuration returns."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, 'tests', 'fakes', 'libfake_rccl.so')


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


FAKE_NOCOUNT = os.path.join(ROOT, 'tests', 'fakes', 'libfake_rccl_nocount.so')


@pytest.fixture(scope='module')
def fake_rccl():
    r = subprocess.run(['make', '-C', os.path.join(ROOT, 'tests', 'fakes')], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and os.path.exists(FAKE) and os.path.exists(FAKE_NOCOUNT), r.stdout
    return FAKE


def _launch(world, script, args, env_extra, timeout=300):
    env = dict(os.environ)
    env.update(env_extra)
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), script] + args
    return subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout, cwd=ROOT)


CASES = [
    # one-workgroup select, safe variant (costs travel with the shard), ragged shard sizes for the tile plan (N / W = 256)
    dict(name='w2_safe', world=2, N=512, H=10, k=51, I=3, variant='safe', seed=83, plan_seed=23, calls=4),
    # three ranks, a population the fused multi-workgroup select takes (N >= 24000), elite set spread over all shards
    dict(name='w3_fused_select', world=3, N=24576, H=6, k=2457, I=2, variant='cem', seed=84, plan_seed=5, calls=3),
    # the 8-launch select chain under the graph, two ranks
    dict(name='w2_chain_select', world=2, N=4096, H=8, k=409, I=3, variant='cem', seed=85, plan_seed=9, calls=3, select_mode=2),
    # the split-product rollout, sharded
    dict(name='w2_split_precision', world=2, N=1024, H=8, k=102, I=3, variant='safe', seed=86, plan_seed=3, calls=3, precision='bf16x3'),
]


# the collective refuses to be captured into a hipGraph (CEM_FAKE_RCCL_CAPTURE=error: ncclAllGather on a capturing stream returns
# ncclInvalidUsage): cem_planner_plan must drop to eager launches for good (cem_capi.hip: graph_failed), report
# 'graph-unsupported', and still return the single-rank result bit for bit on every rank, call after call
CASES += [
    dict(name='w2_capture_refused', world=2, N=512, H=8, k=51, I=3, variant='cem', seed=87, plan_seed=11, calls=4,
         env={'CEM_FAKE_RCCL_CAPTURE': 'error'}, expect_status=['eager'] + ['graph-unsupported'] * 3),
    dict(name='w3_capture_refused_fused_select', world=3, N=24576, H=4, k=2457, I=2, variant='safe', seed=88, plan_seed=12, calls=3,
         env={'CEM_FAKE_RCCL_CAPTURE': 'error'}, expect_status=['eager'] + ['graph-unsupported'] * 2),
]


@pytest.mark.parametrize('case', CASES, ids=[c['name'] for c in CASES])
def test_every_rank_of_a_sharded_plan_equals_the_single_rank_plan(case, fake_rccl, tmp_path):
    from tests import helpers as hp
    # reference first (this process holds the GPU too: world + 1 processes on the card, within the box's limit of 6)
    pb = hp.make_problem(seed=case['seed'])
    _, cfg = hp.configs(pb, N=case['N'], H=case['H'], P=5, E=5, k=case['k'], I=case['I'], variant=case['variant'], post=0.3, noise=0.02,
                        use_graph=True, select_mode=case.get('select_mode', 0), precision=case.get('precision', 'fp32'))
    ref = hp.make_planner(pb, cfg)
    expect = []
    for c in range(case['calls']):
        a, s, it = ref.plan(pb['state'], seed=case['plan_seed'], call=c)
        expect.append((a, np.float32(s), it, ref.mu_sigma().cpu().numpy(), np.sort(ref.elite_idx().cpu().numpy())))
    ref.close()

    r = _launch(case['world'], os.path.join(ROOT, 'tests', 'multirank_worker.py'), [str(tmp_path), json.dumps(case)],
                dict({'CEM_RCCL_LIBRARY': fake_rccl}, **case.get('env', {})))
    assert r.returncode == 0, r.stdout[-4000:]
    assert 'cem_mpc: RCCL entry points bound from CEM_RCCL_LIBRARY=' + fake_rccl in r.stdout      # the override is never silent
    for rank in range(case['world']):
        got = np.load(os.path.join(str(tmp_path), 'rank%d.npz' % rank))
        st = json.load(open(os.path.join(str(tmp_path), 'rank%d.json' % rank)))['graph_status']
        if 'expect_status' in case:
            assert st == case['expect_status'], st
        else:
            assert st[0] == 'eager' and all(x == 'graph' for x in st[1:]), st      # the collective was captured and replayed
        for c in range(case['calls']):
            a, s, it, ms, el = expect[c]
            np.testing.assert_array_equal(got['action%d' % c], a, err_msg='rank %d call %d' % (rank, c))
            assert got['score%d' % c] == s and got['iters%d' % c] == it
            np.testing.assert_array_equal(got['musig%d' % c], ms)
            np.testing.assert_array_equal(got['elite%d' % c], el)
        np.testing.assert_array_equal(got['action_step'], expect[1][0])
        assert got['score_step'] == expect[1][1]


def test_bench_multi_rank_leg_two_ranks_on_one_gpu(fake_rccl):
    """bench.py --gpus 2 as the driver launches it, except that both ranks use cuda:0 (CEM_BENCH_SHARE_GPU=1: gloo carries the
    barrier / max-over-ranks, the fake carries the all-gather).  The line must be the multi-rank one: native exchange, weak
    scaling, the B5 leg present.  (Its numbers mean nothing: two ranks time-share one GPU.)"""
    r = _launch(2, os.path.join(ROOT, 'bench.py'), ['--gpus', '2', '--steps', '3', '--warmup', '2', '--no-cpu-baseline'],
                {'CEM_RCCL_LIBRARY': fake_rccl, 'CEM_BENCH_SHARE_GPU': '1'}, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{') and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-4000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['config']['candidates_per_gpu'] == 2000 and 'N=4000' in d['config']['workload']
    assert d['config']['exchange'].startswith('ncclAllGather inside the library'), d['config']
    assert d['config']['hip_graph'] is True
    # the unscaled rate and the literal metric ("(N=2000 K=5 H=30) at G GPUs": 2000 candidates in total, 1000 per rank) beside the weak-scaled value
    assert abs(d['weak_plans_per_s'] * 2.0 - d['value']) < 1e-6 * d['value'] and d['weak_candidates_per_plan'] == 4000 and 'extras_timed_out' not in d
    assert len(d['per_rank_single_gpu_b2_plans_per_s']) == 2 and min(d['per_rank_single_gpu_b2_plans_per_s']) > 0 and d['order'].startswith('the other legs first')
    b2s = d['b2_strong']
    assert b2s['n_ranks'] == 2 and b2s['candidates_per_rank'] == 1000 and b2s['scaling'] == 'strong' and b2s['unit'] == 'plans/s' and 'N=2000' in b2s['workload']
    assert b2s['exchange'].startswith('ncclAllGather inside the library') and b2s['hip_graph'] is True and b2s['n_ranks_seen_by_rccl'] == 2 and b2s['plans_per_s'] > 0
    assert d['value'] > 0 and d['b5']['n_ranks'] == 2 and d['b5']['candidates_per_rank'] == 32768 and d['b5']['n_ranks_seen_by_rccl'] == 2
    assert d['b5']['exchange'].startswith('ncclAllGather inside the library') and d['b5']['hip_graph'] is True
    sp = d['b5_split_bf16x3']
    assert sp['precision'] == 'bf16x3' and sp['n_ranks'] == 2 and sp['hip_graph'] is True and sp['plans_per_s'] > 0


def _bench_two_ranks(env, extra_args=()):
    r = _launch(2, os.path.join(ROOT, 'bench.py'), ['--gpus', '2', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-b5'] + list(extra_args),
                dict({'CEM_BENCH_SHARE_GPU': '1'}, **env), timeout=600)
    assert r.returncode == 0, r.stdout[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{') and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-4000:]
    return json.loads(lines[0]), r.stdout


def test_bench_refuses_a_communicator_of_the_wrong_size(fake_rccl):
    """ncclCommCount disagrees with --gpus (the fake reports n + 1): bench.py must not time a plan over that communicator — every
    rank sees the mismatch, says so, and ALL ranks fall back together to the host-stepped exchange (no rank is left in a
    collective the others never enter); the line says which exchange produced the number."""
    d, log = _bench_two_ranks({'CEM_RCCL_LIBRARY': fake_rccl, 'CEM_FAKE_RCCL_COUNT_OFFSET': '1'})
    assert 'the RCCL communicator has 3 ranks, --gpus is 2' in log
    assert d['n_gpus'] == 2 and d['config']['exchange'] == 'torch.distributed all_gather between ctypes calls' and d['config']['hip_graph'] is False
    assert d['value'] > 0


def test_bench_with_an_rccl_that_lacks_ncclCommCount(fake_rccl):
    """The optional ncclCommCount export is missing (libfake_rccl_nocount.so): cem_planner_comm_ranks returns CEM_ERR_COMM, the library
    says at load time that the count is unavailable, bench.py reports `not verified` and carries on over the native exchange."""
    d, log = _bench_two_ranks({'CEM_RCCL_LIBRARY': FAKE_NOCOUNT})
    assert 'no ncclCommCount: cem_planner_comm_ranks unavailable' in log and 'communicator rank count not verified' in log
    assert d['n_gpus'] == 2 and d['config']['exchange'].startswith('ncclAllGather inside the library') and d['config']['hip_graph'] is True
