"""A self-contained stand-in for Safety-Gym's Point-Goal tasks (MuJoCo and safety_gym are not in this image): a 2-D point
robot with heading, a goal disc, hazard discs and vase markers, observed through the same kind of sensors the reference
wraps — accelerometer, gyro, magnetometer, velocimeter and 16-bin pseudo-lidars per object class — and delivered through
the same observation post-processing as ``MbrlSafetyGym.fix_observation`` (reference
simba/environment_utils/safety_gym.py:68-93: lidars are flipped to 1 - x so that larger means farther).

Only what the planner path reads is reproduced (observation layout, info['cost'], info['goal_met'], the scorer
constants); the physics is a damped unicycle, not MuJoCo."""
import numpy as np

from .safety_gym import SyntheticSafetyGym


class PointGoalEnv(SyntheticSafetyGym):
    """gym-style reset() / step(action) -> (obs, reward, done, info)."""

    def __init__(self, n_hazards=8, n_vases=1, num_steps=1000, extents=1.5, lidar_bins=16, seed=None, config=None):
        sensors = dict(accelerometer=3, goal_lidar=lidar_bins, gyro=3, hazards_lidar=lidar_bins, magnetometer=3,
                       vases_lidar=lidar_bins, velocimeter=3)
        cfg = dict(lidar_num_bins=lidar_bins)
        cfg.update(config or {})
        super().__init__(sensors=sensors, act_dim=2, config=cfg)
        self.n_hazards, self.n_vases, self.num_steps, self.extents = n_hazards, n_vases, num_steps, extents
        self.rng = np.random.default_rng(seed)
        self.dt = 0.02
        self.reset()

    # ---- world -----------------------------------------------------------------------------------------------------------
    def _place(self, keepout, others):
        for _ in range(1000):
            p = self.rng.uniform(-self.extents, self.extents, 2)
            if all(np.linalg.norm(p - q) > keepout + r for q, r in others):
                return p
        return p

    def _new_goal(self):
        others = [(h, self.config['hazards_size']) for h in self.hazards] + [(self.pos, 0.3)]
        self.goal = self._place(self.config['goal_size'], others)
        self.last_dist_goal = float(np.linalg.norm(self.goal - self.pos))

    def reset(self, **kwargs):
        self.steps = 0
        self.pos = self.rng.uniform(-self.extents, self.extents, 2)
        self.theta = self.rng.uniform(0, 2 * np.pi)
        self.vel, self.omega, self.acc = 0.0, 0.0, 0.0
        self.hazards = []
        for _ in range(self.n_hazards):
            self.hazards.append(self._place(0.18, [(h, 0.2) for h in self.hazards] + [(self.pos, 0.4)]))
        self.vases = [self._place(0.15, [(h, 0.2) for h in self.hazards] + [(self.pos, 0.3)]) for _ in range(self.n_vases)]
        self._new_goal()
        return self._observe()

    # ---- sensors -----------------------------------------------------------------------------------------------------------
    def _lidar(self, positions):
        """Pseudo-lidar: per bin the strongest linear return max(0, D - dist)/D, aliased into both neighbouring bins."""
        bins, D = self.config['lidar_num_bins'], float(self.config['lidar_max_dist'])
        out = np.zeros(bins)
        c, s = np.cos(self.theta), np.sin(self.theta)
        for p in positions:
            d = p - self.pos
            ego = np.array([c * d[0] + s * d[1], -s * d[0] + c * d[1]])
            dist = float(np.hypot(*ego))
            ang = np.arctan2(ego[1], ego[0]) % (2 * np.pi)
            size = 2 * np.pi / bins
            b = int(ang / size) % bins
            sensor = max(0.0, D - dist) / D
            out[b] = max(out[b], sensor)
            alias = (ang - b * size) / size
            out[(b + 1) % bins] = max(out[(b + 1) % bins], alias * sensor)
            out[(b - 1) % bins] = max(out[(b - 1) % bins], (1 - alias) * sensor)
        return out

    def _observe(self):
        t = self.sensor_offset_table
        obs = np.zeros(self.observation_space.shape[0], np.float64)
        obs[t['accelerometer']] = [self.acc, self.vel * self.omega, 9.81 + self.rng.normal(0, 0.01)]
        obs[t['gyro']] = [0.0, 0.0, self.omega]
        obs[t['magnetometer']] = [np.cos(self.theta), -np.sin(self.theta), 0.0]
        obs[t['velocimeter']] = [self.vel, 0.0, 0.0]
        # fix_observation (safety_gym.py:75-86): 1 - lidar
        obs[t['goal_lidar']] = 1.0 - self._lidar([self.goal])
        obs[t['hazards_lidar']] = 1.0 - self._lidar(self.hazards)
        obs[t['vases_lidar']] = 1.0 - self._lidar(self.vases)
        return obs

    # ---- dynamics ------------------------------------------------------------------------------------------------------------
    def step(self, action):
        a = np.clip(np.asarray(action, np.float64), -1.0, 1.0)
        self.acc = 3.0 * a[0] - 2.0 * self.vel
        self.vel += self.dt * self.acc
        self.omega = 3.0 * a[1]
        self.theta = (self.theta + self.dt * self.omega) % (2 * np.pi)
        self.pos = np.clip(self.pos + self.dt * self.vel * np.array([np.cos(self.theta), np.sin(self.theta)]),
                           -self.extents - 0.5, self.extents + 0.5)
        self.steps += 1
        dist = float(np.linalg.norm(self.goal - self.pos))
        reward = (self.last_dist_goal - dist) * self.config['reward_distance']
        self.last_dist_goal = dist
        info = {}
        if dist <= self.config['goal_size']:
            reward += self.config['reward_goal']
            info['goal_met'] = True
            self._new_goal()
        cost = float(any(np.linalg.norm(h - self.pos) <= self.config['hazards_size'] for h in self.hazards))
        info['cost'] = cost
        info['cost_hazards'] = cost
        return self._observe(), float(reward), self.steps >= self.num_steps, info
