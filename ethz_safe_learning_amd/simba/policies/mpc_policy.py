"""MpcPolicy, reference simba/policies/mpc_policy.py:8-57: holds model / reward source / action space / H / N / P.

The reference's ``compute_objective`` (:26-39) is a TF op chain over a materialised trajectory tensor; here it is the
epilogue of the fused rollout kernel (csrc/cem_device.h), so the method is not exposed as a separate op."""
from ..spaces import Box
from ...planner import sampling_params
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

    def compute_objective(self, trajectories, action_sequences):
        raise NotImplementedError('compute_objective is fused into the rollout kernel (cem_rollout_kernel epilogue); '
                                  'use CemMpc.generate_action or CemPlanner.plan_rollout + scores_local()')

    def build(self):
        pass

    @property
    def sampling_params(self):
        """lower_bound, upper_bound, mean, stddev (mpc_policy.py:45-57)."""
        return sampling_params(self.action_space.low, self.action_space.high)
