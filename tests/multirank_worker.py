"""One rank of a candidate-sharded plan on a box whose ranks all share cuda:0 (tests/test_gpu_multirank.py starts W of these under
torch.distributed.run).  torch.distributed (gloo) only carries the communicator id; the collective inside the plan is the
library's ncclAllGather call, bound to tests/fakes/libfake_rccl.so through CEM_RCCL_LIBRARY.  Each rank writes what it saw to
<out>/rank<r>.npz; the test compares all ranks with a single-rank planner of the same configuration."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, case_json = sys.argv[1], sys.argv[2]
    case = json.loads(case_json)
    import torch
    import torch.distributed as dist
    from tests import helpers as hp
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo')
    torch.cuda.set_device(0)
    pb = hp.make_problem(seed=case['seed'])
    _, cfg = hp.configs(pb, N=case['N'], H=case['H'], P=5, E=5, k=case['k'], I=case['I'], variant=case['variant'], post=0.3, noise=0.02,
                        use_graph=True, world_size=world, rank=rank, select_mode=case.get('select_mode', 0), precision=case.get('precision', 'fp32'))
    pl = hp.make_planner(pb, cfg)
    pl.comm_init()
    assert pl.comm_ranks() == world
    res = {}
    status = []
    for c in range(case['calls']):                     # call 0 eager, call 1 captures the graph (collective included), later calls replay
        a, s, it = pl.plan(pb['state'], seed=case['plan_seed'], call=c)
        res['action%d' % c] = a
        res['score%d' % c] = np.float32(s)
        res['iters%d' % c] = np.int32(it)
        res['musig%d' % c] = pl.mu_sigma().cpu().numpy()
        res['elite%d' % c] = np.sort(pl.elite_idx().cpu().numpy())
        status.append(pl.graph_status())
    # the stepwise form with the library's exchange
    pl.plan_begin(pb['state'], seed=case['plan_seed'], call=1)
    for it in range(case['I']):
        pl.plan_rollout(it)
        pl.plan_exchange()
        pl.plan_select(it)
    a, s, it = pl.plan_end()
    res['action_step'] = a
    res['score_step'] = np.float32(s)
    dist.barrier()
    pl.comm_destroy()
    pl.close()
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), **res)
    with open(os.path.join(out_dir, 'rank%d.json' % rank), 'w') as f:
        json.dump({'graph_status': status}, f)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
