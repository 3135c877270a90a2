"""End to end through the reference's plugin contract: scripts/train.py -> make_environment -> make_agent
(eval-by-name policy / model construction) -> RLTrainer.train: random warm-up, MlpEnsemble.fit on the GPU, SafeCemMpc
planning on the GPU through Policy.generate_action, evaluation reports."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_script_smoke(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    import train as train_script
    trainer = train_script.main(['--config_dir', os.path.join(ROOT, 'ethz_safe_learning_amd', 'config'), '--config_basename', 'smoke.yaml',
                                 '--log_dir', str(tmp_path), '--name', 'smoke', '--seed', '3', '--log_level', 'WARNING'])
    agent = trainer.agent
    assert agent.warm and agent.total_training_steps >= 1200 + 2 * 300
    assert agent.model.model._trainer.iterations == 3 * 400            # one fit per iteration, Adam state persistent
    assert agent.policy._planner is not None and agent.policy.last_iterations >= 1
    runs = [d for d in os.listdir(tmp_path) if d.startswith('smoke_')]
    assert len(runs) == 1 and os.path.exists(os.path.join(tmp_path, runs[0], 'params.txt'))
    recs = [json.loads(l) for l in open(os.path.join(tmp_path, runs[0], 'training_data', 'scalars.jsonl'))]
    tags = {r['tag'] for r in recs}
    assert {'eval_rl_objective', 'sum_rewards_stddev', 'eval_mean_sum_costs', 'sum_costs', 'training_rl_objective', 'mean_sum_costs'} <= tags
    assert all(np.isfinite(r['value']) for r in recs)
