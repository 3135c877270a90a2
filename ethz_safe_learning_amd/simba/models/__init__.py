from .model import BaseModel                        # noqa: F401
from .mlp_ensemble import MlpEnsemble               # noqa: F401
from .transition_model import TransitionModel       # noqa: F401
