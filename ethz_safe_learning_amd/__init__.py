"""MI355X-native CEM-MPC planner: the ensemble-rollout hot path of
yardenas/ethz-safe-learning ("simba") as fused gfx950 HIP kernels behind a C ABI
(include/cem_mpc.h), with a simba-shaped Python host (``.simba``).

Importing this package never touches the GPU and never imports ``oracle/``.
"""
from .planner import CemPlanner, PlannerConfig, ScorerConfig, flatten_weights, pack_weights_host, plan_tiles, sampling_params  # noqa: F401

__all__ = ['CemPlanner', 'PlannerConfig', 'ScorerConfig', 'flatten_weights', 'pack_weights_host', 'plan_tiles',
           'sampling_params']
