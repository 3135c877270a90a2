"""MbrlAgent: random warm-up, then model-based control; every update fits the transition model on the most recent
transitions with goal-met steps masked out (interface of reference simba/agents/mbrl_agent.py:9-118).  Policy and model
classes are looked up by the CamelCase of their YAML names, exactly as the reference does (:103-118)."""
import numpy as np

from ..infrastructure.common import standardize_name
from ..infrastructure.logging_utils import logger
from ..models.transition_model import TransitionModel
from ..policies import CemMpc, RandomMpc, RandomShootingMpc, SafeCemMpc            # noqa: F401  (resolved by name)
from .agent import BaseAgent

_POLICIES = dict(CemMpc=CemMpc, SafeCemMpc=SafeCemMpc, RandomMpc=RandomMpc, RandomShootingMpc=RandomShootingMpc)


class MbrlAgent(BaseAgent):
    def __init__(self, environment, warmup_timesteps, train_batch_size, train_interaction_steps, episode_length,
                 replay_buffer_size, **kwargs):
        super().__init__(replay_buffer_size, **kwargs)
        for key in ('policy', 'policy_params', 'model', 'model_params'):
            assert key in kwargs, "Did not specify a policy or a model."
        self.observation_space_dim = environment.observation_space.shape[0]
        self.actions_space_dim = environment.action_space.shape[0]
        self.train_batch_size = train_batch_size
        self.train_interaction_steps = train_interaction_steps
        self.episode_length = episode_length
        self.warmup_timesteps = warmup_timesteps
        self.total_warmup_timesteps_so_far = 0
        self.warmup_policy = self._make_policy('random_mpc', kwargs['policy_params'], environment)
        model_params = dict(kwargs['model_params'], scale_features=kwargs['scale_features'])
        self.model = self._make_model(kwargs['model'], model_params, environment, kwargs['sampling_propagation'])
        self.policy = self._make_policy(kwargs['policy'], kwargs['policy_params'], environment)

    @property
    def warm(self):
        return self.total_warmup_timesteps_so_far >= self.warmup_timesteps

    def update(self):
        obs, acts, next_obs, _, _, infos = self.replay_buffer.sample_recent_data(self.train_batch_size)
        # transitions on which the goal was met are discontinuous (the goal is re-sampled): keep them out of the fit
        keep = ~np.array([bool(info.get('goal_met', False)) for info in infos])
        self.model.fit(np.concatenate([obs[keep], acts[keep]], axis=1), next_obs[keep])

    def _interact(self, environment):
        if not self.warm:
            samples, steps = self.sample_trajectories(environment, self.warmup_policy, self.warmup_timesteps, self.episode_length)
            self.total_warmup_timesteps_so_far += steps
            return samples, steps
        return self.sample_trajectories(environment, self.policy, self.train_interaction_steps, self.episode_length)

    def _build(self):
        self.model.build()
        self.policy.build()
        logger.info('Done building Mbrl agent computational graph.')

    def _load(self):
        raise NotImplementedError

    def report(self, environment, eval_interaction_steps, eval_episode_length):
        logger.info('Evaluating policy.')
        trajectories, _ = self.sample_trajectories(environment, self.policy, eval_interaction_steps, eval_episode_length)
        returns = np.array([tr['reward'].sum() for tr in trajectories])
        costs = np.array([sum(info.get('cost', 0.0) for info in tr['info']) for tr in trajectories])
        self.training_report.update(eval_rl_objective=returns.mean(), sum_rewards_stddev=returns.std(), eval_mean_sum_costs=costs.mean())
        return self.training_report

    def _make_policy(self, policy, policy_params, environment):
        cls = _POLICIES[standardize_name(policy)]
        if cls is RandomMpc:
            return RandomMpc(environment.action_space)
        return cls(model=self.model, **dict(policy_params, environment=environment))

    def _make_model(self, model, model_params, environment, sampling_propagation):
        return TransitionModel(model=model, observation_space=environment.observation_space, action_space=environment.action_space,
                               sampling_propagation=sampling_propagation, **model_params)
