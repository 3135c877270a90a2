"""Experiment configuration: built-in defaults for policies / models / agents (the hyper-parameters the reference ships
in config/policies.yaml, config/models.yaml, config/agents.yaml) overridden recursively by an experiment YAML
(reference config/config.py:5-39).  ``load_config_or_die(config_dir, config_basename)`` keeps the reference's signature;
if the directory also holds models.yaml / policies.yaml / agents.yaml they replace the built-in defaults, and the
reference's experiment names (experiment, experiment_no_sample, experiment_unaware, debug, tune_policy) resolve to built-in
presets when no file of that name exists."""
import copy
import os

import yaml

# key order as the reference's loader produces it (models.yaml, policies.yaml, agents.yaml, then the experiment file's `options`):
# pretty_print — what scripts/train.py dumps into params.txt — shows it
DEFAULTS = {
    'models': {
        'mlp_ensemble': dict(ensemble_size=15, batch_size=64, validation_split=0.2, learning_rate=0.00025, learning_rate_schedule=True,
                             training_steps=5000, mlp_params=dict(n_layers=4, units=128, activation='tf.nn.relu', dropout_rate=0.0)),
    },
    'policies': {
        'cem_mpc': dict(horizon=8, iterations=10, smoothing=0.0, n_samples=150, n_elite=15, particles=5, stddev_threshold=0.25,
                        noise_stddev=0.001),
        'safe_cem_mpc': dict(horizon=8, iterations=9, smoothing=0.0, n_samples=500, n_elite=20, particles=45, stddev_threshold=0.25,
                             noise_stddev=0.01, posterior_mean_threashold=0.15),
        'random_shooting_mpc': dict(horizon=10, n_samples=1000),      # (config/policies.yaml:21-23; the class itself cannot be constructed)
    },
    'agents': {
        'agent': dict(replay_buffer_size=1000000, action_repeat=6, add_observation_noise=False),
        'mbrl_agent': dict(train_batch_size=30000, train_interaction_steps=1000, episode_length=1000, warmup_timesteps=5000,
                           policy='safe_cem_mpc', model='mlp_ensemble', scale_features=True, sampling_propagation=True),
    },
}


def _options(log_frequency, eval_steps, eval_len, train_iterations, fps=60, seed=None):
    o = dict(trainer_options=dict(video_log_frequency=-1, log_frequency=log_frequency, max_video_length=1000,
                                  eval_interaction_steps=eval_steps, eval_episode_length=eval_len,
                                  training_logger_params=dict(fps=fps)))
    if seed is not None:
        o['seed'] = seed                      # (config/tune_policy.yaml:10: between trainer_options and train_iterations)
    o.update(train_iterations=train_iterations, agent='mbrl_agent', environment='MbrlSafexp-PointSimpleGoal1-v0')
    return o


# The experiment set the reference ships (config/experiment*.yaml, debug.yaml, tune_policy.yaml; scripts/run_experiments.sh
# selects them with --config_basename), as overrides of DEFAULTS.  A YAML of that name in --config_dir takes precedence.
PRESETS = {
    'experiment': dict(options=_options(7, 4000, 1000, 125)),
    'experiment_no_sample': dict(options=_options(7, 4000, 1000, 125), agents=dict(mbrl_agent=dict(sampling_propagation=False))),
    'experiment_unaware': dict(options=_options(7, 4000, 1000, 125), agents=dict(mbrl_agent=dict(policy='cem_mpc'))),
    'debug': dict(options=_options(1, 25, 25, 100)),
    'tune_policy': dict(options=_options(-1, 25, 25, 60, fps=28, seed=1), models=dict(mlp_ensemble=dict(ensemble_size=5))),
}


def overwrite_default_values(update_from, update_to):
    for key, value in update_from.items():
        if isinstance(value, dict) and isinstance(update_to.get(key), dict):
            overwrite_default_values(value, update_to[key])
        else:
            update_to[key] = value
    return update_to


def load_config_or_die(config_dir, config_basename):
    config = copy.deepcopy(DEFAULTS)
    for filename in ('models.yaml', 'policies.yaml', 'agents.yaml'):
        path = os.path.join(config_dir, filename)
        if os.path.exists(path):
            with open(path, 'r') as fh:
                config.update(yaml.safe_load(fh))
    path = os.path.join(config_dir, config_basename)
    if not os.path.exists(path) and os.path.exists(path + '.yaml'):
        path += '.yaml'
    if os.path.exists(path):
        with open(path, 'r') as fh:
            overwrite_default_values(yaml.safe_load(fh) or {}, config)
    else:
        preset = os.path.splitext(os.path.basename(config_basename))[0]
        if preset not in PRESETS:
            raise FileNotFoundError('no experiment file %s and no built-in experiment %r (built-ins: %s)'
                                    % (path, preset, ', '.join(sorted(PRESETS))))
        overwrite_default_values(copy.deepcopy(PRESETS[preset]), config)
    return config


def pretty_print(config, indent=0):
    lines = []
    for key, value in config.items():
        head = '  ' * indent + str(key).ljust(30 - 2 * indent)
        if isinstance(value, dict):
            lines.append(head)
            lines.append(pretty_print(value, indent + 1).rstrip('\n'))
        else:
            lines.append(head + str(value))
    return '\n'.join(lines) + '\n'
