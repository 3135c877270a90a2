"""simba-shaped host API over the HIP planner: the module paths, class names and constructor kwargs of
yardenas/ethz-safe-learning's plugin interface for the CEM-MPC path, so that
``MbrlAgent._make_policy`` / ``_make_model`` (reference simba/agents/mbrl_agent.py:103-118) resolve to these
classes unchanged.  Built: the hot path (section 8a) and the rows section 8f marks "next" — ensemble training on the device, the agent /
trainer / replay-buffer / config harness, the shape-keyed planner cache, the remaining 'goal'-task scorer branches.  Not built, and
raising where the reference itself cannot run: the 'push' task scorer and `random_shooting_mpc`."""
