"""simba-shaped host API over the HIP planner: the module paths, class names and constructor kwargs of
yardenas/ethz-safe-learning's plugin interface for the CEM-MPC path, so that
``MbrlAgent._make_policy`` / ``_make_model`` (reference simba/agents/mbrl_agent.py:103-118) resolve to these
classes unchanged.  Only the hot path is implemented; what section 8 marks "next" raises NotImplementedError."""
