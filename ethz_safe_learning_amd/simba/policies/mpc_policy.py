"""MpcPolicy, reference simba/policies/mpc_policy.py:8-57: holds model / reward source / action space / H / N / P.

The reference's ``compute_objective`` (:26-39) is a TF op chain over a materialised trajectory tensor.  Inside
``generate_action`` it is the epilogue of the fused rollout kernel (csrc/cem_device.h) and no trajectory exists; as a
method of its own it runs ``cem_compute_objective`` (the same arithmetic as a standalone HBM-bound kernel) on the
trajectory tensor the caller passes."""
import numpy as np

from ..spaces import Box
from ...planner import PlannerConfig, cached_planner, sampling_params
from .policy import PolicyBase


class MpcPolicy(PolicyBase):
    def __init__(self, model, environment, horizon, n_samples, particles):
        super().__init__()
        self.model = model
        self.environment = environment
        self.reward = getattr(environment, 'get_reward', None)
        self.action_space = environment.action_space
        assert isinstance(self.action_space, Box) or all(hasattr(self.action_space, a) for a in ('low', 'high', 'shape')), \
            "Expecting only box as action space."
        self.horizon = horizon
        self.n_samples = n_samples
        self.particles = particles

    def generate_action(self, state):
        raise NotImplementedError

    variant = 'cem'              # 'safe' in SafeCemMpc: which compute_objective this class has

    def _objective_extra_config(self):
        return {}

    def _objective_planner(self):
        """A handle that carries this policy's objective (variant, particles, scorer): the planning handle of a built
        CemMpc, otherwise a minimal one (its sampling shape is irrelevant to compute_objective)."""
        pl = getattr(self, '_planner', None)
        if pl is not None and pl.h is not None:
            return pl
        scorer = getattr(self.environment, '_scorer', None) or getattr(self.environment, 'scorer', None)
        if scorer is None:
            raise ValueError('environment must expose its SafetyGymStateScorer as `_scorer` (safety_gym.py:27-29)')
        m, ens = self.model, self.model.model
        cfg = PlannerConfig(obs_dim=m.observation_space_dim, act_dim=m.action_space_dim, ensemble_size=ens.ensemble_size,
                            particles=self.particles, n_samples=ens.ensemble_size, horizon=1, n_elite=1, iterations=1,
                            scorer=scorer.to_scorer_config(), act_low=self.action_space.low, act_high=self.action_space.high,
                            units=ens.mlp_params['units'], n_layers=ens.mlp_params['n_layers'], activation=ens.activation, variant=self.variant,
                            **self._objective_extra_config())
        return cached_planner(cfg, device=getattr(self, 'device', 'cuda:0'))

    def compute_objective(self, trajectories, action_sequences=None):
        """mpc_policy.py:26-39 (SafeCemMpc: safe_cem_mpc.py:76-96): trajectories [particles*n, H+1, obs] in the tf.tile row
        order of cem_mpc.py:49-51 -> scores [n].  ``action_sequences`` is accepted for signature parity; get_reward /
        get_cost ignore actions (safety_gym.py:62-66).  numpy in -> numpy out, torch in -> torch (GPU) out."""
        scores = self._objective_planner().compute_objective(trajectories)
        return scores.cpu().numpy() if isinstance(trajectories, np.ndarray) else scores

    def build(self):
        pass

    @property
    def sampling_params(self):
        """lower_bound, upper_bound, mean, stddev (mpc_policy.py:45-57)."""
        return sampling_params(self.action_space.low, self.action_space.high)
