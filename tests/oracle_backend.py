"""An oracle-backed stand-in for CemPlanner's stepwise interface, used ONLY by the CPU tests of the sharded driver
(tests may use the oracle; the product never does).  Rank g evaluates candidates [g*N/G, (g+1)*N/G) x all particles
with the members their GLOBAL rows select, exactly as the HIP tiles do (csrc/cem_capi.hip build_plan_tiles)."""
import numpy as np
import torch

from oracle import cem_oracle as o


class OracleBackend:
    def __init__(self, pb, ocfg, world_size, rank):
        self.pb, self.cfg, self.W, self.R = pb, ocfg, world_size, rank
        N = ocfg.n_samples
        self.nloc = N // world_size
        self._scores_local = torch.zeros(self.nloc, dtype=torch.float32)
        self._scores_global = torch.zeros(N, dtype=torch.float32) if world_size > 1 else self._scores_local
        self.trace = []

    def scores_local(self):
        return self._scores_local

    def scores_global(self):
        return self._scores_global

    def plan_begin(self, state, seed=0, call=0, eps_act=None, eps_model=None):
        self.state = np.asarray(state, np.float32)
        self.eps_act, self.eps_model = eps_act, eps_model
        A = self.pb['low'].shape[0]
        self.lb, self.ub, mu0, sg0 = o.sampling_params(self.pb['low'], self.pb['high'])
        self.mu = np.broadcast_to(mu0, (self.cfg.horizon, A)).astype(np.float32).copy()
        self.sigma = np.broadcast_to(sg0, (self.cfg.horizon, A)).astype(np.float32).copy()
        self.best, self.best_score, self.iters, self.done = np.zeros(A, np.float32), np.float32(-np.inf), 0, False
        self.trace = []

    def plan_rollout(self, it):
        if self.done:
            return
        cfg, N, P = self.cfg, self.cfg.n_samples, self.cfg.particles
        self.actions = o.sample_actions(self.mu, self.sigma, self.lb, self.ub, self.eps_act[it])      # replicated sampling
        n0, n1 = self.R * self.nloc, (self.R + 1) * self.nloc
        rows = np.concatenate([p * N + np.arange(n0, n1) for p in range(P)])
        members = o.member_of_rows(P * N, cfg.ensemble_size, rows)
        sc = o.candidate_scores(self.state, self.actions[n0:n1], self.pb['weights'], self.pb['inputs_min'], self.pb['inputs_max'],
                                self.eps_model[it][:, rows], cfg, self.pb['scorer'], members=members)
        self._scores_local.copy_(torch.from_numpy(sc.astype(np.float32)))

    def plan_select(self, it):
        if self.done:
            return
        scores = self._scores_global.numpy().copy()
        self.mu, self.sigma, self.best, self.best_score, elite, stop = o.select_and_refit(
            scores, self.actions, self.mu, self.sigma, self.best, self.best_score, self.cfg)
        self.iters += 1
        self.done = stop
        self.trace.append(dict(scores=scores, elite=elite, mu=self.mu.copy(), sigma=self.sigma.copy()))

    def plan_end(self, eps_out=None):
        a = self.best + (np.asarray(eps_out, np.float32) if eps_out is not None else 0) * np.float32(self.cfg.noise_stddev)
        return a, float(self.best_score), self.iters
