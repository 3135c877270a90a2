"""Candidate-sharded CEM plan: one process per GPU, one exchange step per CEM
iteration.

Rank g owns candidates [g*N/G, (g+1)*N/G) x all particles (so the particle mean
and the Beta count stay local, mpc_policy.py:38-39 / safe_cem_mpc.py:111-112).
Action sampling is replicated (Philox keyed on the GLOBAL candidate index, or
the same explicit eps_act tensor on every rank), so after ONE all-gather of the
per-candidate scores every rank runs the identical top-k / moments refit and
holds identical mu, sigma, best-so-far: results are bit-identical to the
single-GPU plan for any world size.  The early-stop flag is a function of the
replicated sigma, so ranks agree without further traffic (cem_mpc.py:66-67).

``backend`` is anything with plan_begin / plan_rollout / plan_select / plan_end
and scores_local() / scores_global() tensors; in production it is a
``CemPlanner`` (HIP).  The driver itself never computes.

This host-stepped driver is the portable form (any torch.distributed backend:
the gloo test ranks on CPU use it).  On GPUs the preferred form is
``CemPlanner.comm_init()`` + ``CemPlanner.plan()``: the library then owns an RCCL
communicator and runs the same loop — all-gather included — natively, as one
hipGraph per rank.
"""
from __future__ import annotations


class ShardedCemDriver:
    def __init__(self, backend, iterations, world_size=1, group=None, always_exchange=False):
        self.backend = backend
        self.iterations = iterations
        self.world_size = world_size
        self.group = group
        self.always_exchange = always_exchange      # issue the collective even for one rank (exercises RCCL on a one-GPU box)

    def exchange(self):
        if getattr(self.backend, 'has_comm', False):
            self.backend.plan_exchange()              # the handle's own RCCL communicator (cem_plan_exchange): no torch in the loop
            return
        if self.world_size == 1 and not self.always_exchange:
            return
        import contextlib
        import torch.distributed as dist
        # payload: N/G floats per rank (B5: 32 KB) -> latency bound; RCCL picks its one-shot small-message path.
        # The collective is issued with the planner's stream current, so it is ordered after the rollout/reduce
        # kernels and before the select kernel without any host synchronisation.
        if hasattr(self.backend, 'stream_context'):       # CemPlanner: stream order holds, the accessors need not drain the stream
            with self.backend.stream_context():
                dist.all_gather_into_tensor(self.backend.scores_global(sync=False), self.backend.scores_local(sync=False), group=self.group)
            return
        with contextlib.nullcontext():
            dist.all_gather_into_tensor(self.backend.scores_global(), self.backend.scores_local(), group=self.group)

    def plan(self, state, seed=0, call=0, eps_act=None, eps_model=None, eps_out=None):
        b = self.backend
        b.plan_begin(state, seed=seed, call=call, eps_act=eps_act, eps_model=eps_model)
        for it in range(self.iterations):
            b.plan_rollout(it)
            self.exchange()
            b.plan_select(it)
        return b.plan_end(eps_out=eps_out)
