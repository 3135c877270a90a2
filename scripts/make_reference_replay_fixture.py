"""Generates tests/golden/reference_replay_buffer.json FROM THE REFERENCE'S OWN MODULE: loads
/root/reference/simba/infrastructure/replay_buffer.py by file path (NumPy only — with config/config.py the other module of the
reference that needs no TensorFlow; its package __init__ pulls tensorboardX, so it is not imported as a package) and
drives it through a scripted sequence:

  * path_summary on Python lists (float64 / int inputs -> float32 arrays, infos passed through),
  * concatenate_rollouts,
  * ReplayBuffer(max_size=50, add_noise=False): four store() calls of 7 + 12, 20, 30 and 60 transitions (the third crosses max_size, the
    fourth alone exceeds it), after each: the six arrays, sample_recent_data(10), sample_recent_rollouts(2) lengths, and
    sample_random_data(8) under np.random.seed(1000 + call),
  * add_noise() on its own under np.random.seed(7) on data with an all-zero column (the `mean == 0 -> 1e-5` rule),
  * ReplayBuffer(max_size=40, add_noise=True) under np.random.seed(11): two store() calls.

Runs only in the build container (the reference does not travel to the GPU box); only the JSON — inputs and the outputs the
reference computed — is committed.  tests/test_harness_cpu.py drives this repo's ReplayBuffer through the same script and holds it
to the fixture array for array (values, dtype, shape).  `sample_random_rollouts` is left out: the reference's
`np.array(self.paths, copy=False)` raises under NumPy >= 2 (the NumPy of this image).

    python scripts/make_reference_replay_fixture.py [/root/reference]
"""
import importlib.util
import json
import os
import sys

import numpy as np


def make_paths(rng, lengths, obs_dim=5, act_dim=2):
    """Scripted rollouts as plain Python lists (what BaseAgent.sample_trajectory hands to path_summary, agent.py:147-153)."""
    paths = []
    for n in lengths:
        obs = rng.normal(0.0, 1.0, (n, obs_dim)).round(4)
        obs[:, 3] = 0.0                                             # an all-zero column: add_noise's 1e-5 rule
        nxt = (obs + rng.normal(0.0, 0.1, (n, obs_dim))).round(4)
        nxt[:, 3] = 0.0
        acts = rng.uniform(-1, 1, (n, act_dim)).round(4)
        rews = rng.normal(0.0, 1.0, n).round(4)
        terms = [0] * (n - 1) + [1]                                 # ints, as the environment's `done` flags
        infos = [dict(cost=float(rng.integers(0, 2)), goal_met=bool(rng.integers(0, 2))) for _ in range(n)]
        paths.append(dict(observations=obs.tolist(), actions=acts.tolist(), rewards=rews.tolist(), next_observations=nxt.tolist(),
                          terminals=terms, infos=infos))
    return paths


def enc(a):
    a = np.asarray(a)
    if a.dtype == object:
        return dict(dtype='object', shape=list(a.shape), data=a.tolist())
    return dict(dtype=str(a.dtype), shape=list(a.shape), data=a.astype(np.float64).tolist())


SIX = ('observations', 'actions', 'next_observations', 'terminals', 'rewards', 'infos')


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
    src = os.path.join(ref, 'simba', 'infrastructure', 'replay_buffer.py')
    spec = importlib.util.spec_from_file_location('reference_replay_buffer', src)
    rb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rb)

    rng = np.random.default_rng(2024)
    batches = [make_paths(rng, [7, 12]), make_paths(rng, [20]), make_paths(rng, [30]), make_paths(rng, [60])]
    out = dict(generated_by='scripts/make_reference_replay_fixture.py',
               source='simba/infrastructure/replay_buffer.py:4-116 of the reference (loaded by file path, NumPy %s)' % np.__version__,
               batches=batches, steps=[])

    def summarise(raw):
        return [rb.path_summary(p['observations'], p['actions'], p['rewards'], p['next_observations'], p['terminals'], p['infos']) for p in raw]

    # path_summary / concatenate_rollouts on the first batch
    ps = summarise(batches[0])
    out['path_summary'] = [{k: (enc(v) if k != 'info' else v) for k, v in p.items()} for p in ps]
    out['concatenate_rollouts'] = [enc(a) for a in rb.concatenate_rollouts(ps)]

    buf = rb.ReplayBuffer(50, False)
    for i, raw in enumerate(batches):
        buf.store(summarise(raw))
        step = dict(arrays={k: enc(getattr(buf, k)) for k in SIX},
                    sample_recent_data=[enc(a) for a in buf.sample_recent_data(10)],
                    recent_rollout_lengths=[int(p['observation'].shape[0]) for p in buf.sample_recent_rollouts(2)],
                    n_paths=len(buf.paths))
        np.random.seed(1000 + i)
        step['sample_random_data'] = [enc(a) for a in buf.sample_random_data(8)]
        out['steps'].append(step)

    np.random.seed(7)
    data = np.asarray(batches[1][0]['observations'], np.float32)
    out['add_noise'] = dict(seed=7, data=enc(data), result=enc(rb.add_noise(data.copy())), result_nts_0p05=None)
    np.random.seed(8)
    out['add_noise']['result_nts_0p05'] = enc(rb.add_noise(data.copy(), noise_to_signal=0.05))

    np.random.seed(11)
    nbuf = rb.ReplayBuffer(40, True)
    noisy = []
    for raw in batches[:2]:
        nbuf.store(summarise(raw))
        noisy.append({k: enc(getattr(nbuf, k)) for k in SIX})
    out['noisy_buffer'] = dict(seed=11, max_size=40, steps=noisy)

    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(here, 'tests', 'golden', 'reference_replay_buffer.json')
    with open(path, 'w') as fh:
        json.dump(out, fh)
        fh.write('\n')
    print('wrote %s (%d bytes)' % (path, os.path.getsize(path)))


if __name__ == '__main__':
    main()
