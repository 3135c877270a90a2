from .safety_gym import SafetyGymStateScorer, SyntheticSafetyGym     # noqa: F401
