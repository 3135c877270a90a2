#!/usr/bin/env python3
"""One rank's plan time at B2 with the exchange step in its three forms (one GPU = one RCCL rank, so the all-gather is the
1-rank in-place case; what is compared is the launch structure around it):
  host-stepped   torch.distributed all_gather_into_tensor between ctypes calls (sharded.ShardedCemDriver)
  native eager   ncclAllGather issued by the library on the planner stream, kernel by kernel
  native graph   the same, captured once into one hipGraph per rank and replayed
Run under torch.distributed.run --nproc-per-node 1."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from ethz_safe_learning_amd import CemPlanner, PlannerConfig, synthetic
from ethz_safe_learning_amd.sharded import ShardedCemDriver
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
pb = synthetic.problem(60, 2, 5)


def planner(graph):
    cfg = PlannerConfig(obs_dim=60, act_dim=2, ensemble_size=5, particles=5, n_samples=2000, horizon=30, n_elite=200, iterations=5,
                        scorer=pb['scorer'], act_low=pb['low'], act_high=pb['high'], noise_stddev=1e-3, use_graph=graph)
    pl = CemPlanner(cfg); pl.set_weights(pb['weights']); pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
    return pl


def timed(fn, n=60):
    for i in range(6):
        fn(i)
    torch.cuda.synchronize(); t = []
    for i in range(n):
        t0 = time.perf_counter(); fn(100 + i); t.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(t))


res = {}
pl = planner(False); drv = ShardedCemDriver(pl, 5, world_size=1, always_exchange=True)
res['host_stepped_ms'] = timed(lambda i: drv.plan(pb['state'], seed=1, call=i))
pl = planner(False); pl.comm_init()
res['native_eager_ms'] = timed(lambda i: pl.plan(pb['state'], seed=1, call=i)); res['native_eager_status'] = pl.graph_status()
pl = planner(True); pl.comm_init()
res['native_graph_ms'] = timed(lambda i: pl.plan(pb['state'], seed=1, call=i)); res['native_graph_status'] = pl.graph_status()
pl = planner(True)
res['single_rank_graph_no_collective_ms'] = timed(lambda i: pl.plan(pb['state'], seed=1, call=i))
print(json.dumps(res))
dist.destroy_process_group()
