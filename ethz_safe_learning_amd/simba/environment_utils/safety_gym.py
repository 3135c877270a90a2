"""The scorer side of reference simba/environment_utils/safety_gym.py.

``SafetyGymStateScorer`` keeps the reference's constructor (config dict copied onto attributes by setattr +
sensor_offset_table, :104-108) and turns it into the parameter block of the fused reward/cost epilogue
(``to_scorer_config``).  Inside the planner its arithmetic (:110-192) is the epilogue of cem_rollout_kernel; ``reward`` /
``cost`` as methods of their own run the same arithmetic through cem_scorer_reward / cem_scorer_cost (HIP; no host scorer).

``SyntheticSafetyGym`` stands in for ``MbrlSafetyGym`` (:9-101), whose MuJoCo / safety_gym simulator is not in this
image: it provides the attributes the planner path reads — observation_space, action_space, sensor_offset_table,
_scorer, get_reward / get_cost."""
import numpy as np

from ..spaces import Box
from ...planner import ScorerConfig

# safety_gym Engine.DEFAULT values the scorer reads (upstream package, absent here; SURVEY 8a end) overridden by the
# reference's registry for Simple-Goal tasks (safety_gym_registery.py:9-16,27-40)
ENGINE_DEFAULTS = dict(task='goal', goal_size=0.3, hazards_size=0.2, vases_size=0.1, pillars_size=0.2, gremlins_size=0.1,
                       lidar_max_dist=4, lidar_num_bins=5, observe_goal_lidar=True, observe_goal_dist=False,
                       constrain_hazards=True, constrain_vases=False, constrain_pillars=False, constrain_gremlins=False,
                       constrain_indicator=True, reward_distance=1.0, reward_goal=1.0, reward_clip=10,
                       reward_orientation=False)


class SafetyGymStateScorer(object):
    def __init__(self, config, sensor_offset_table):
        for key, value in config.items():
            setattr(self, key, value)
        self.sensor_offset_table = sensor_offset_table

    @staticmethod
    def _bounds(sl):
        return (int(sl.start), int(sl.stop)) if isinstance(sl, slice) else (int(sl[0]), int(sl[1]))

    def to_scorer_config(self):
        if getattr(self, 'task', 'goal') != 'goal':
            raise NotImplementedError("only task 'goal' is built (the 'push' branch, safety_gym.py:120-134, is SURVEY 8f-4)")
        if getattr(self, 'reward_orientation', False):
            raise NotImplementedError('reward_orientation indexes a non-existent sensor in the reference (safety_gym.py:136-139)')
        if getattr(self, 'observe_goal_lidar', False):
            goal, lidar = self._bounds(self.sensor_offset_table['goal_lidar']), True
        elif getattr(self, 'observe_goal_dist', False):
            goal, lidar = self._bounds(self.sensor_offset_table['goal_dist']), False
        else:
            raise NotImplementedError                                   # safety_gym.py:175-176
        kinds = []
        for kind in ('vases', 'hazards', 'pillars', 'gremlins'):        # evaluation order of safety_gym.py:148-163
            if getattr(self, 'constrain_' + kind, False):
                lo, hi = self._bounds(self.sensor_offset_table[kind + '_lidar'])
                kinds.append((lo, hi, float(getattr(self, kind + '_size'))))
        return ScorerConfig(goal_slice=goal, observe_goal_lidar=lidar, lidar_max_dist=float(self.lidar_max_dist),
                            goal_size=float(self.goal_size), reward_distance=float(self.reward_distance),
                            reward_goal=float(self.reward_goal), reward_clip=float(self.reward_clip or 0.0),
                            constrain_indicator=bool(self.constrain_indicator), cost_kinds=kinds)

    def _handle(self, device='cuda:0'):
        """A planner handle that carries only this scorer (the scorer ops read nothing else from it)."""
        from ...planner import PlannerConfig, cached_planner
        obs_dim = max(self._bounds(sl)[1] for sl in self.sensor_offset_table.values())
        cfg = PlannerConfig(obs_dim=obs_dim, act_dim=1, ensemble_size=1, particles=1, n_samples=1, horizon=1, n_elite=1,
                            iterations=1, scorer=self.to_scorer_config(), act_low=[-1.0], act_high=[1.0])
        return cached_planner(cfg, device=device)

    def reward(self, observations, next_observations):
        """safety_gym.py:110-119,140-143 ('goal' task): (reward [n], goal_achieved [n]).  numpy in -> numpy out."""
        r, g = self._handle().scorer_reward(observations, next_observations)
        return (r.cpu().numpy(), g.cpu().numpy()) if isinstance(observations, np.ndarray) else (r, g)

    def cost(self, observations):
        """safety_gym.py:145-166: cost [n]."""
        c = self._handle().scorer_cost(observations)
        return c.cpu().numpy() if isinstance(observations, np.ndarray) else c


class SyntheticSafetyGym(object):
    """Observation layout of a Safety-Gym task without the simulator: sorted sensor keys -> contiguous slices
    (safety_gym.py:17-25), lidars bounded [0,1], everything else unbounded (:34-60)."""

    def __init__(self, sensors=None, act_dim=2, config=None):
        if sensors is None:      # stock PointGoal1: accelerometer, goal_lidar, gyro, hazards_lidar, magnetometer, vases_lidar, velocimeter
            sensors = dict(accelerometer=3, goal_lidar=16, gyro=3, hazards_lidar=16, magnetometer=3, vases_lidar=16, velocimeter=3)
        self.sensor_offset_table = {}
        low, high, offset = [], [], 0
        for k, size in sorted(sensors.items()):
            self.sensor_offset_table[k] = slice(offset, offset + size)
            bounded = k.endswith('_lidar') or k == 'remaining'
            low += [0.0 if bounded else -np.inf] * size
            high += [1.0 if bounded else np.inf] * size
            offset += size
        self.observation_space = Box(np.asarray(low), np.asarray(high), dtype=np.float32)
        self.action_space = Box(-np.ones(act_dim), np.ones(act_dim), dtype=np.float32)
        cfg = dict(ENGINE_DEFAULTS)
        cfg.update(config or {})
        self.config = cfg
        self._scorer = SafetyGymStateScorer(cfg, self.sensor_offset_table)

    def get_reward(self, obs, acs, *args, **kwargs):
        return self._scorer.reward(obs, *args, **kwargs)

    def get_cost(self, obs, acs, *args, **kwargs):
        return self._scorer.cost(obs)
