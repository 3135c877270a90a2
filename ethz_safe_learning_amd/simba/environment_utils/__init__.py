from .safety_gym import SafetyGymStateScorer, SyntheticSafetyGym     # noqa: F401
from .point_goal_env import PointGoalEnv                             # noqa: F401
from .environment_factory import make_environment                    # noqa: F401
