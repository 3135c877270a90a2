"""CemMpc, reference simba/policies/cem_mpc.py:6-68, on the HIP planner.

Same constructor kwargs (cem_mpc.py:7-17), same ``generate_action(state) -> np.float32[A]`` contract
(cem_mpc.py:31-33; caller simba/agents/agent.py:120).  One ``CemPlanner`` handle corresponds to the reference's one
traced ``@tf.function`` graph; it is built lazily on the first call and rebuilt never (a shape change is a new
policy object, as in scripts/tune_cem_policy.py:109-115).  Weights / normaliser are re-staged whenever the model's
``version`` changed (after ``fit``: mbrl_agent.py:53)."""
import numpy as np

from ...planner import PlannerConfig, cached_planner
from .mpc_policy import MpcPolicy


class CemMpc(MpcPolicy):
    variant = 'cem'

    def __init__(self, model, environment, horizon, iterations, smoothing, n_samples, n_elite, particles,
                 stddev_threshold, noise_stddev, seed=0, device='cuda:0', use_graph=True, precision='fp32'):
        super().__init__(model, environment, horizon, n_samples, particles)
        self.iterations = iterations
        self.smoothing = smoothing
        self.elite = n_elite
        self.stddev_threshold = stddev_threshold
        self.noise_stddev = noise_stddev
        self.seed = seed
        self.device = device
        self.use_graph = use_graph
        self.precision = precision                     # 'fp32' | 'bf16x3' (PlannerConfig.precision; beyond the reference's kwargs)
        self._planner = None
        self.last_score = None
        self.last_iterations = None

    # ---- planner plumbing -------------------------------------------------------------------------------------
    def _extra_config(self):
        return {}

    def _scorer_config(self):
        scorer = getattr(self.environment, '_scorer', None) or getattr(self.environment, 'scorer', None)
        if scorer is None:
            raise ValueError('environment must expose its SafetyGymStateScorer as `_scorer` (as MbrlSafetyGym does, '
                             'reference simba/environment_utils/safety_gym.py:27-29)')
        return scorer.to_scorer_config()

    def planner_config(self):
        m = self.model
        ens = m.model
        return PlannerConfig(
            obs_dim=m.observation_space_dim, act_dim=m.action_space_dim, ensemble_size=ens.ensemble_size,
            particles=self.particles, n_samples=self.n_samples, horizon=self.horizon, n_elite=self.elite,
            iterations=self.iterations, scorer=self._scorer_config(), act_low=self.action_space.low,
            act_high=self.action_space.high, units=ens.mlp_params['units'], n_layers=ens.mlp_params['n_layers'], activation=ens.activation,
            smoothing=self.smoothing, stddev_threshold=self.stddev_threshold, noise_stddev=self.noise_stddev,
            variant=self.variant, sampling_propagation=m.sampling_propagation, scale_features=m.scale_features,
            use_graph=self.use_graph, precision=self.precision, **self._extra_config())

    def build(self):
        if self._planner is None or self._planner.h is None:      # never built, or closed by its owner
            # one handle per distinct shape, shared by every policy object of that shape (tune_cem_policy.py:109-115)
            self._planner = cached_planner(self.planner_config(), device=self.device)
        self._sync_model()

    def _sync_model(self):
        tag = (self.model.uid, self.model.version)            # uid, not id(): ids are reused after garbage collection
        if self._planner.staged != tag:
            self._planner.set_weights(self.model.model.get_weights())
            self._planner.set_normaliser(self.model.inputs_min, self.model.inputs_max)
            self._planner.staged = tag

    # ---- the plugin boundary ------------------------------------------------------------------------------------
    def generate_action(self, state):
        self.build()                                   # cached handle + weights of the current model version
        action, score, iters = self._planner.plan(np.asarray(state, np.float32), seed=self.seed)
        self.last_score, self.last_iterations = score, iters
        return action

    def do_generate_action(self, state, eps_act=None, eps_model=None, eps_out=None):
        """(action, best_score) like cem_mpc.py:35-68; explicit noise tensors replace TF's stateful RNG."""
        self.build()
        action, score, iters = self._planner.plan(np.asarray(state, np.float32), seed=self.seed, eps_act=eps_act,
                                                  eps_model=eps_model, eps_out=eps_out)
        self.last_iterations = iters
        return action, score
