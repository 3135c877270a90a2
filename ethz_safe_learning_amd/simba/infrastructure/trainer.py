"""Outer training loop: interact -> update -> (periodically) report, reference simba/infrastructure/trainer.py:7-93.
Constructor keywords are the ``trainer_options`` of the experiment YAML (incl. the reference's spelling ``environemnt``)."""
import numpy as np

from .logging_utils import TrainingLogger, logger


def _returns_and_costs(trajectories):
    returns = np.array([float(np.sum(tr['reward'])) for tr in trajectories])
    costs = np.array([float(sum(step_info.get('cost', 0.0) for step_info in tr['info'])) for tr in trajectories])
    return returns, costs


class RLTrainer(object):
    def __init__(self, agent, environemnt, log_frequency, video_log_frequency, max_video_length, eval_interaction_steps,
                 eval_episode_length, training_logger_params):
        self.agent = agent
        self.environment = environemnt
        self.training_logger = TrainingLogger(**training_logger_params)
        self.log_frequency = log_frequency
        self.video_log_frequency = video_log_frequency
        self.max_video_length = max_video_length
        self.eval_interaction_steps = eval_interaction_steps
        self.eval_episode_length = eval_episode_length

    def train(self, iterations):
        self.agent.build_graph()
        for iteration in range(iterations):
            logger.info('Training iteration %d.', iteration)
            self.agent.interact(self.environment)
            self.agent.update()
            if self.log_frequency > 0 and iteration % self.log_frequency == 0:
                self.log(self.agent.report(self.environment, self.eval_interaction_steps, self.eval_episode_length), iteration)
            if self.video_log_frequency > 0 and iteration % self.video_log_frequency == 0:
                self.log_video(self.agent.render_trajectory(environment=self.environment, policy=self.agent.policy,
                                                            max_trajectory_length=self.max_video_length), iteration)

    def evaluate_agent(self, interaction_steps, max_trajectory_length):
        trajectories, _ = self.agent.sample_trajectories(self.environment, self.agent.policy, interaction_steps, max_trajectory_length)
        returns, costs = _returns_and_costs(trajectories)
        return dict(training_rl_objective=returns.mean(), sum_rewards_stddev=returns.std(), sum_costs_mean=costs.mean(),
                    sum_costs_stddev=costs.std())

    def log(self, report, epoch):
        """Scalars of one report (trainer.py:65-82): training return mean / stddev, mean episode cost + whatever the agent added."""
        report = dict(report)
        returns, costs = _returns_and_costs(report.pop('training_trajectories'))
        report.update(training_rl_objective=returns.mean(), sum_rewards_stddev=returns.std(), mean_sum_costs=costs.mean())
        step = report.pop('total_training_steps')
        for key, value in report.items():
            self.training_logger.log_scalar(value, key, step)
        self.training_logger.flush()

    def log_video(self, trajectory_records, epoch):
        self.training_logger.log_video(trajectory_records, 'what_the_policy_looks_like', epoch)
