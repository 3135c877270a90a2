#!/usr/bin/env python3
"""Counterpart of the reference's scripts/tune_cem_policy.py (:56-137): train, then grid-search the CEM planner over
horizon x (proposals, iterations) x elite ratio by swapping ``agent.policy`` for fresh ``CemMpc`` objects and evaluating
each.  Every distinct (H, I, N, k) is a new planner handle (the reference re-traces its tf.function); handles are cached
by shape.  Results go to <log_dir>/grid_search.json (and scores.svg / costs.svg when matplotlib is installed).

  python scripts/tune_cem_policy.py --config_dir ethz_safe_learning_amd/config --config_basename smoke.yaml \
         --eval_steps 300 --eval_episode_length 300
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))

HORIZONS = [8, 10, 12, 15]
PROPOSALS_WITH_ITERATIONS = [(100, 15), (150, 10), (300, 5)]
ELITE_RATIOS = [0.05, 0.1, 0.2]


def make_new_policy(model, environment, horizon, iterations, n_samples, elite_ratio, policy_kwargs):
    from ethz_safe_learning_amd.simba.policies import CemMpc
    return CemMpc(model=model, environment=environment, horizon=horizon, iterations=iterations, n_samples=n_samples,
                  n_elite=round(elite_ratio * n_samples), particles=policy_kwargs['particles'],
                  stddev_threshold=policy_kwargs['stddev_threshold'], noise_stddev=policy_kwargs['noise_stddev'],
                  smoothing=policy_kwargs['smoothing'], seed=policy_kwargs.get('seed', 0))


def grid_search(trainer, env, params, eval_steps, eval_episode_length, horizons=HORIZONS,
                proposals_with_iterations=PROPOSALS_WITH_ITERATIONS, elite_ratios=ELITE_RATIOS):
    from ethz_safe_learning_amd.simba.infrastructure.logging_utils import logger
    agent = trainer.agent
    results = []
    for horizon in horizons:
        for n_samples, iterations in proposals_with_iterations:
            for ratio in elite_ratios:
                agent.policy = make_new_policy(agent.model, env, horizon, iterations, n_samples, ratio, params['policies']['cem_mpc'])
                t0 = time.perf_counter()
                m = trainer.evaluate_agent(eval_steps, eval_episode_length)
                rec = dict(horizon=horizon, n_samples=n_samples, iterations=iterations, elite_ratio=ratio,
                           n_elite=agent.policy.elite, score_mean=float(m['training_rl_objective']),
                           score_std=float(m['sum_rewards_stddev']), cost_mean=float(m['sum_costs_mean']),
                           cost_std=float(m['sum_costs_stddev']), seconds=time.perf_counter() - t0)
                logger.info('H=%d (N,I)=(%d,%d) elite %.2f: score %.3f +- %.3f, cost %.3f +- %.3f', horizon, n_samples, iterations,
                            ratio, rec['score_mean'], rec['score_std'], rec['cost_mean'], rec['cost_std'])
                results.append(rec)
    return results


def main(argv=None):
    import train as train_script
    ap = argparse.ArgumentParser()
    ap.add_argument('--name', type=str, default='')
    ap.add_argument('--log_dir', type=str, default='experiments')
    ap.add_argument('--log_level', type=str, default='INFO')
    ap.add_argument('--config_dir', type=str, required=True)
    ap.add_argument('--config_basename', type=str, required=True)
    ap.add_argument('--cuda_device', type=str, default='0')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--eval_steps', type=int, default=7000)                 # tune_cem_policy.py:116
    ap.add_argument('--eval_episode_length', type=int, default=1000)
    ap.add_argument('--quick', action='store_true', help='2 x 2 x 2 corner of the grid (tests)')
    args = ap.parse_args(argv)
    from ethz_safe_learning_amd.config.config import load_config_or_die
    params = load_config_or_die(args.config_dir, args.config_basename)
    trainer = train_script.main(['--config_dir', args.config_dir, '--config_basename', args.config_basename, '--log_dir', args.log_dir,
                                 '--name', args.name, '--seed', str(args.seed), '--log_level', args.log_level,
                                 '--cuda_device', args.cuda_device])
    grid = dict(horizons=HORIZONS[:2], proposals_with_iterations=PROPOSALS_WITH_ITERATIONS[1:], elite_ratios=ELITE_RATIOS[:2]) if args.quick else {}
    results = grid_search(trainer, trainer.environment, params, args.eval_steps, args.eval_episode_length, **grid)
    out_dir = trainer.training_logger.log_dir or args.log_dir
    with open(os.path.join(out_dir, 'grid_search.json'), 'w') as fh:
        json.dump(results, fh, indent=1)
    try:
        import matplotlib
        matplotlib.use('Agg')
        import matplotlib.pyplot as plt
        for key, cmap in (('score_mean', 'Blues'), ('cost_mean', 'Reds')):
            hs = sorted({r['horizon'] for r in results})
            fig, axes = plt.subplots(1, len(hs), sharey='all', figsize=(3 * len(hs), 3))
            for ax, h in zip(np.atleast_1d(axes), hs):
                rows = [r for r in results if r['horizon'] == h]
                ne = len({r['elite_ratio'] for r in rows})
                ax.pcolor(np.array([r[key] for r in rows]).reshape(-1, ne), cmap=cmap)
                ax.set_title('H=%d' % h)
            fig.savefig(os.path.join(out_dir, key + '.svg'))
    except ImportError:
        pass
    return results


if __name__ == '__main__':
    main()
