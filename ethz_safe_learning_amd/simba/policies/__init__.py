from .policy import PolicyBase                      # noqa: F401
from .mpc_policy import MpcPolicy                   # noqa: F401
from .cem_mpc import CemMpc                         # noqa: F401
from .safe_cem_mpc import SafeCemMpc                # noqa: F401
from .random_mpc import RandomMpc                   # noqa: F401
from .random_shooting_mpc import RandomShootingMpc   # noqa: F401
