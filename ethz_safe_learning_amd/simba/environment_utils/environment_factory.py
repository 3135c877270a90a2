"""make_environment(config) (reference simba/environment_utils/environment_factory.py:6-13).  The reference builds
Safety-Gym / MuJoCo tasks; here every 'MbrlSafexp-Point*Goal*' name maps to the self-contained PointGoalEnv stand-in
(level 0: no hazards, level 1: 8 hazards + 1 vase, level 2: 10 hazards + 10 vases)."""
import re

from .point_goal_env import PointGoalEnv


def make_environment(config, seed=None):
    name = config['options']['environment']
    m = re.match(r'^(Mbrl)?Safexp-Point(Simple)?Goal([012])-v0$', name)
    if not m:
        raise ValueError('no synthetic stand-in for environment %r (MuJoCo / safety_gym are not available)' % name)
    hazards, vases = {'0': (0, 0), '1': (8, 1), '2': (10, 10)}[m.group(3)]
    return PointGoalEnv(n_hazards=hazards, n_vases=vases, seed=seed,
                        config=dict(constrain_hazards=hazards > 0, **(config['options'].get('environment_params') or {})))
