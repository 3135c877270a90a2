from .agent import BaseAgent                         # noqa: F401
from .mbrl_agent import MbrlAgent                    # noqa: F401
from .agent_factory import make_agent                # noqa: F401
