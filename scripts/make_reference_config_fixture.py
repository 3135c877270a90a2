"""Generates tests/golden/reference_configs.json FROM THE REFERENCE'S OWN LOADER: imports /root/reference/config/config.py
(PyYAML only — the one module of the reference that is importable without TensorFlow) and dumps what its
`load_config_or_die` returns, and what its `pretty_print` prints, for the five experiment files the reference ships
(config/experiment.yaml, experiment_unaware.yaml, experiment_no_sample.yaml, debug.yaml, tune_policy.yaml).

Runs only in the build container (the reference does not travel to the GPU box); only the JSON — data the reference computed — is
committed.  tests/test_harness_cpu.py holds this repo's config loader and its built-in presets against it key for key.

    python scripts/make_reference_config_fixture.py [/root/reference]
"""
import importlib.util
import json
import os
import sys

EXPERIMENTS = ['experiment', 'experiment_unaware', 'experiment_no_sample', 'debug', 'tune_policy']


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
    cfg_dir = os.path.join(ref, 'config')
    spec = importlib.util.spec_from_file_location('reference_config', os.path.join(cfg_dir, 'config.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = dict(generated_by='scripts/make_reference_config_fixture.py', source='config/config.py:5-39 + config/*.yaml of the reference',
               experiments={})
    for name in EXPERIMENTS:
        cfg = mod.load_config_or_die(cfg_dir, name + '.yaml')
        out['experiments'][name] = dict(config=cfg, pretty_print=mod.pretty_print(cfg))
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(here, 'tests', 'golden', 'reference_configs.json')
    with open(path, 'w') as fh:
        json.dump(out, fh, indent=1)          # insertion order kept: the loader's key order is part of what pretty_print shows
        fh.write('\n')
    print('wrote %s (%d experiments)' % (path, len(EXPERIMENTS)))


if __name__ == '__main__':
    main()
