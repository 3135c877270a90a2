#!/usr/bin/env python3
"""Counterpart of the reference's scripts/train.py (:14-45) on this repo's MI355X planner: same flags, config merge,
seeding, params.txt dump, make_environment -> make_agent -> RLTrainer(**trainer_options).train(train_iterations).

  python scripts/train.py --config_dir ethz_safe_learning_amd/config --config_basename smoke.yaml --name demo
"""
import argparse
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    from ethz_safe_learning_amd.config.config import load_config_or_die, pretty_print
    from ethz_safe_learning_amd.simba.agents.agent_factory import make_agent
    from ethz_safe_learning_amd.simba.environment_utils.environment_factory import make_environment
    from ethz_safe_learning_amd.simba.infrastructure.logging_utils import init_logging, logger
    from ethz_safe_learning_amd.simba.infrastructure.trainer import RLTrainer
    ap = argparse.ArgumentParser()
    ap.add_argument('--name', type=str, default='')
    ap.add_argument('--log_dir', type=str, default='experiments')
    ap.add_argument('--log_level', type=str, default='INFO')
    ap.add_argument('--config_dir', type=str, required=True)
    ap.add_argument('--config_basename', type=str, required=True)
    ap.add_argument('--cuda_device', type=str)
    ap.add_argument('--seed', type=int, default=1)
    args = ap.parse_args(argv)
    if args.cuda_device is not None:                     # the same knob, spelled for ROCm as well
        os.environ['CUDA_VISIBLE_DEVICES'] = args.cuda_device
        os.environ['HIP_VISIBLE_DEVICES'] = args.cuda_device
    log_dir = os.path.join(args.log_dir, args.name + '_' + time.strftime('%d-%m-%Y_%H-%M-%S'))
    os.makedirs(log_dir, exist_ok=True)
    init_logging(args.log_level)
    params = load_config_or_die(args.config_dir, args.config_basename)
    np.random.seed(args.seed)
    logger.info('Starting a training session with parameters:\n' + pretty_print(params))
    try:
        git_hash = subprocess.check_output(['git', 'rev-parse', 'HEAD'], cwd=ROOT, stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        git_hash = 'unknown'
    with open(os.path.join(log_dir, 'params.txt'), 'w') as fh:
        fh.write(pretty_print(params) + '\ngit hash: ' + git_hash)
    env = make_environment(params, seed=args.seed)
    for p in params['policies'].values():
        p.setdefault('seed', args.seed)                  # the planner's Philox key (TF's global seed in the reference)
    agent = make_agent(params, env)
    trainer_options = params['options'].pop('trainer_options')
    trainer_options['training_logger_params'].update(log_dir=os.path.join(log_dir, 'training_data'))
    trainer = RLTrainer(agent=agent, environemnt=env, **trainer_options)
    trainer.train(params['options'].pop('train_iterations'))
    return trainer


if __name__ == '__main__':
    main()
