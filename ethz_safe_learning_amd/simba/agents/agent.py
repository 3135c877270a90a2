"""BaseAgent: collects environment rollouts with action repeat and keeps the replay buffer and cost bookkeeping
(interface of reference simba/agents/agent.py:7-153).  The planner is called once per ``action_repeat`` environment
steps with a NumPy observation and must return a NumPy action of the action space's shape (agent.py:120,146)."""
import numpy as np

from ..infrastructure import replay_buffer as rb
from ..infrastructure.logging_utils import logger


class BaseAgent(object):
    def __init__(self, replay_buffer_size, add_observation_noise, action_repeat, *args, **kwargs):
        assert action_repeat, "Action repeat should be at least 1."
        self.replay_buffer = rb.ReplayBuffer(replay_buffer_size, add_observation_noise)
        self.action_repeat = action_repeat
        self.training_report = dict()
        self.total_training_steps = 0

    # ---- the trainer-facing protocol ----------------------------------------------------------------------------------
    def interact(self, environment):
        samples, steps = self._interact(environment)
        self.total_training_steps += steps
        self.replay_buffer.store(samples)
        batch_cost = sum(float(info.get('cost', 0.0)) for tr in samples for info in tr['info'])
        self.training_report['sum_costs'] = self.training_report.get('sum_costs', 0.0) + batch_cost
        self.training_report['training_trajectories'] = samples
        self.training_report['total_training_steps'] = self.total_training_steps

    def update(self):
        raise NotImplementedError

    def _interact(self, environment):
        raise NotImplementedError

    def build_graph(self, graph_dir=None):
        if graph_dir is None:
            logger.info('Building computational graph.')
            self._build()
        else:
            logger.info('Loading computational graph from %s', graph_dir)
            self._load()

    def _build(self):
        raise NotImplementedError

    def _load(self):
        raise NotImplementedError

    def report(self, environment, eval_interaction_steps, eval_episode_length):
        return self.training_report

    def render_trajectory(self, environment, policy, max_trajectory_length):
        raise NotImplementedError('the synthetic environments have no renderer (MuJoCo is not in this image)')

    # ---- rollouts -------------------------------------------------------------------------------------------------------
    def sample_trajectories(self, environment, policy, batch_size, max_trajectory_length):
        trajectories, steps = [], 0
        while steps < batch_size:
            trajectory, length = self.sample_trajectory(environment, policy, max_trajectory_length)
            trajectories.append(trajectory)
            steps += length
        return trajectories, steps

    def sample_trajectory(self, environment, policy, max_trajectory_length, pbar=None):
        """One episode.  Each decision is held for ``action_repeat`` simulator steps; rewards and costs of the held
        steps are summed into one transition; the hold is cut short when the goal is met or the episode ends
        (agent.py:119-143)."""
        observation = environment.reset()
        rec = dict(o=[], a=[], r=[], o2=[], d=[], info=[])
        steps, over = 0, False
        while not over:
            action = policy.generate_action(observation)
            rec['o'].append(observation)
            rec['a'].append(action)
            held_reward, held_cost, info = 0.0, 0.0, {}
            for _ in range(self.action_repeat):
                observation, reward, done, info = environment.step(action)
                steps += 1
                held_reward += reward
                held_cost += info.get('cost', 0.0)
                over = done or steps == max_trajectory_length
                if over or info.get('goal_met', False):
                    break
            info = dict(info, cost=held_cost)
            rec['o2'].append(observation)
            rec['r'].append(held_reward)
            rec['d'].append(over)
            rec['info'].append(info)
        assert np.shape(rec['a'][0]) == environment.action_space.shape, "Policy produces wrong actions shape."
        return rb.path_summary(rec['o'], rec['a'], rec['r'], rec['o2'], rec['d'], rec['info']), steps
