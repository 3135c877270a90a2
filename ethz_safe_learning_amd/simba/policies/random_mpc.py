"""Warm-up policy, reference simba/policies/random_mpc.py:6-16 (host-side NumPy in the reference too)."""
import numpy as np

from .policy import PolicyBase


class RandomMpc(PolicyBase):
    def __init__(self, action_space):
        super().__init__()
        self.action_space = action_space

    def generate_action(self, state):
        return np.random.uniform(self.action_space.low, self.action_space.high)

    def build(self):
        pass
