"""TransitionModel, reference simba/models/transition_model.py:7-93: the planner-facing dynamics model.
Constructor kwargs as :8-21 (``model`` = 'mlp_ensemble', spaces, scale_features, sampling_propagation, then the
models.yaml keys + train_epochs).  ``unfold_sequences`` / ``simulate_trajectories`` / ``predict`` run the fused HIP
rollout kernel (debug instantiation that writes the trajectory); ``scale`` / ``_fit_statistics`` are the host-side
statistics of :42-50,79-87 that feed the kernel's normaliser."""
import numpy as np

from ..infrastructure.common import standardize_name
from .model import BaseModel
import itertools

from .mlp_ensemble import MlpEnsemble

_MODEL_UIDS = itertools.count(1)

_MODELS = {'MlpEnsemble': MlpEnsemble}


class TransitionModel(BaseModel):
    def __init__(self, model, observation_space, action_space, scale_features, sampling_propagation, **kwargs):
        super().__init__(observation_space.shape[0] + action_space.shape[0], observation_space.shape[0])
        self.model_scope = model
        self.model = _MODELS[standardize_name(model)](inputs_dim=self.inputs_dim, outputs_dim=self.outputs_dim, **kwargs)
        self.observation_space = observation_space
        self.action_space = action_space
        self.scale_features = scale_features
        self.sampling_propagation = sampling_propagation
        self.observation_space_dim = observation_space.shape[0]
        self.action_space_dim = action_space.shape[0]
        self.inputs_min = np.concatenate([observation_space.low, action_space.low]).astype(np.float32)    # :28
        self.inputs_max = np.concatenate([observation_space.high, action_space.high]).astype(np.float32)  # :29
        self._stats_version = 0
        self.uid = next(_MODEL_UIDS)             # process-unique (id() is reused after garbage collection): keys staged weights
        self._planner = None
        self._planner_version = None
        self.seed = 0
        self._calls = 0                          # every unfold / predict without explicit noise draws fresh Philox noise

    @property
    def version(self):
        return (self.model.version, self._stats_version)

    def build(self):
        self.model.build()

    def fit(self, inputs, targets):
        self._fit_statistics(inputs)
        observations = inputs[:, :self.observation_space_dim]
        return self.model.fit(self.scale(np.asarray(inputs, np.float32)), (targets - observations).astype(np.float32))

    def _fit_statistics(self, inputs):
        """transition_model.py:42-50: space bounds where finite, else the data min / max of this fit batch."""
        if not self.scale_features:
            return
        high = np.concatenate([self.observation_space.high, self.action_space.high])
        low = np.concatenate([self.observation_space.low, self.action_space.low])
        self.inputs_min = np.where(np.isfinite(low), low, inputs.min(axis=0)).astype(np.float32)
        self.inputs_max = np.where(np.isfinite(high), high, inputs.max(axis=0)).astype(np.float32)
        self._stats_version += 1

    def scale(self, inputs):
        """transition_model.py:79-87 (host copy for fit(); the rollout kernel applies the same rule on device)."""
        if not self.scale_features:
            return inputs
        delta = self.inputs_max - self.inputs_min
        delta = np.where(delta < np.float32(1e-5), np.float32(1.01), delta)
        return (inputs - self.inputs_min) / delta

    # ---- device rollouts ----------------------------------------------------------------------------------------
    def _get_planner(self, device='cuda:0'):
        from ...planner import CemPlanner, PlannerConfig, ScorerConfig
        if self._planner is None:
            ens = self.model
            cfg = PlannerConfig(obs_dim=self.observation_space_dim, act_dim=self.action_space_dim,
                                ensemble_size=ens.ensemble_size, particles=ens.ensemble_size, n_samples=16, horizon=1,
                                n_elite=1, iterations=1, scorer=ScorerConfig(goal_slice=(0, 1)),
                                act_low=self.action_space.low, act_high=self.action_space.high,
                                units=ens.mlp_params['units'], n_layers=ens.mlp_params['n_layers'], activation=ens.activation,
                                sampling_propagation=self.sampling_propagation, scale_features=self.scale_features)
            self._planner = CemPlanner(cfg, device=device)
        if self._planner_version != self.version:
            self._planner.set_weights(self.model.get_weights())
            self._planner.set_normaliser(self.inputs_min, self.inputs_max)
            self._planner_version = self.version
        return self._planner

    def unfold_sequences(self, s_0, action_sequences, eps_model=None, seed=None, call=None):
        """transition_model.py:64-77: s_0 [B,O], actions [B,H,A] -> trajectories [B,H+1,O] (torch tensor on the GPU).
        Row r is evaluated by member r // (B/E) (mlp_ensemble.py:123-126).  Like the reference's Normal.sample()
        (mlp_ensemble.py:189-193) every call draws fresh noise: the Philox stream is keyed on (self.seed, a per-model call
        counter) unless ``seed`` / ``call`` (or the explicit ``eps_model`` tensor) pin it."""
        if call is None:
            call = self._calls
            self._calls += 1
        return self._get_planner().unfold_sequences(s_0, action_sequences, eps_model=eps_model,
                                                    seed=self.seed if seed is None else seed, call=call)

    def simulate_trajectories(self, current_state, action_sequences, **kw):
        return self.unfold_sequences(current_state, action_sequences, **kw).cpu().numpy()      # :57-61

    def predict(self, inputs, **kw):
        inputs = np.asarray(inputs, np.float32)                                                 # :51-55
        return self.simulate_trajectories(inputs[..., :self.observation_space_dim],
                                          np.expand_dims(inputs[..., -self.action_space_dim:], axis=1), **kw)

    def save(self):
        pass

    def load(self):
        pass
