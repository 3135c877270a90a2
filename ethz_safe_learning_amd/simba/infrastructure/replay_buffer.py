"""Transition store of the agent (interface of reference simba/infrastructure/replay_buffer.py:4-116): rollouts go in as
path dictionaries, training reads the most recent transitions as flat arrays.  Host-side NumPy, as in the reference; unlike
the reference (which re-concatenates every array on each store) chunks are appended and flattened lazily on read."""
import numpy as np

_FIELDS = ('observation', 'action', 'next_observation', 'terminal', 'reward', 'info')


def path_summary(observations, actions, rewards, next_observations, terminals, infos):
    """One rollout as a dictionary of arrays (replay_buffer.py:69-85)."""
    f32 = lambda v: np.asarray(v, dtype=np.float32)          # noqa: E731
    return dict(observation=f32(observations), reward=f32(rewards), action=f32(actions), next_observation=f32(next_observations),
                terminal=f32(terminals), info=infos)


def concatenate_rollouts(paths):
    """-> observations, actions, next_observations, terminals, rewards, infos (replay_buffer.py:88-100)."""
    cat = lambda key: np.concatenate([p[key] for p in paths])      # noqa: E731
    return tuple(cat(k) for k in _FIELDS)


def add_noise(data, noise_to_signal=0.01):
    """Gaussian noise with a per-dimension stddev of noise_to_signal * |mean| (replay_buffer.py:103-116)."""
    scale = np.mean(data, axis=0)
    scale[scale == 0] = 1e-5
    return (data + np.random.normal(0.0, np.abs(scale * noise_to_signal), data.shape)).astype(np.float32)


class ReplayBuffer(object):
    def __init__(self, max_size, add_noise):
        self.max_size = max_size
        self.add_noise = add_noise
        self.paths = []
        self._chunks = {k: [] for k in _FIELDS}
        self._count = 0

    def __len__(self):
        return min(self._count, self.max_size)

    def store(self, paths):
        self.paths.extend(paths)
        arrays = dict(zip(_FIELDS, concatenate_rollouts(paths)))
        if self.add_noise:
            arrays['observation'] = add_noise(arrays['observation'])
            arrays['next_observation'] = add_noise(arrays['next_observation'])
        for k in _FIELDS:
            self._chunks[k].append(arrays[k])
        self._count += arrays['observation'].shape[0]
        if self._count > 2 * self.max_size:                  # amortised trim to the newest max_size transitions
            self._compact()

    def _compact(self):
        for k in _FIELDS:
            self._chunks[k] = [np.concatenate(self._chunks[k])[-self.max_size:]]
        self._count = self._chunks['observation'][0].shape[0]

    def _flat(self, key):
        if len(self._chunks[key]) != 1:
            self._chunks[key] = [np.concatenate(self._chunks[key])]
        return self._chunks[key][0][-self.max_size:]

    # properties with the reference's attribute names
    observations = property(lambda self: self._flat('observation'))
    actions = property(lambda self: self._flat('action'))
    next_observations = property(lambda self: self._flat('next_observation'))
    terminals = property(lambda self: self._flat('terminal'))
    rewards = property(lambda self: self._flat('reward'))
    infos = property(lambda self: self._flat('info'))

    def _take(self, idx):
        return tuple(self._flat(k)[idx] for k in _FIELDS)

    def sample_recent_data(self, batch_size):
        return self._take(slice(-batch_size, None))

    def sample_random_data(self, batch_size):
        return self._take(np.random.permutation(len(self))[-batch_size:])

    def sample_recent_rollouts(self, num_rollouts=1):
        return self.paths[-num_rollouts:]

    def sample_random_rollouts(self, num_rollouts):
        pick = np.random.permutation(len(self.paths))[:num_rollouts]
        return concatenate_rollouts([self.paths[i] for i in pick])
