"""RandomShootingMpc: the `random_shooting_mpc` key of the reference's config/policies.yaml:21.

The reference class (simba/policies/random_shooting_mpc.py:6-39) cannot be constructed or traced: it passes six positional
arguments to MpcPolicy's five-parameter constructor (:14-21 vs mpc_policy.py:8-13), calls tf.random.uniform with the bounds
in the shape position (:29) and uses `self.objective`, which nothing defines.  It therefore has no behaviour to reproduce
(SURVEY.md section 2, row 15), and it is not on the accelerated path.  The name resolves here so that a YAML naming it fails
with this explanation instead of a KeyError."""
from .mpc_policy import MpcPolicy


class RandomShootingMpc(MpcPolicy):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "random_shooting_mpc is not executable in the reference (simba/policies/random_shooting_mpc.py:14-21 raises a "
            "TypeError on construction), so there is nothing to match; use cem_mpc with iterations=1 for one-shot shooting")
