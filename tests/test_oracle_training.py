"""KATs for the oracle's training restatement: the manual backward pass against finite differences, the
Keras Adam update with clipvalue, the per-epoch linear learning-rate decay."""
import numpy as np

from oracle import cem_oracle as o


def _tiny(seed=0, activation='relu'):
    pb = o.synthetic_problem(obs_dim=5, act_dim=2, ensemble_size=2, units=8, n_layers=2, seed=seed, head_scale=0.5, var_bias=-1.0,
                             activation=activation)
    rng = np.random.default_rng(seed)
    w = o.cast_weights(pb['weights'], np.float64)
    for m in w:
        for b in m['b']:
            b[:] = rng.normal(0, 0.1, b.shape)
    x = rng.normal(0, 1, (2, 6, 7))
    y = rng.normal(0, 0.3, (2, 6, 5))
    return w, x, y


import pytest


@pytest.mark.parametrize('activation', ['relu', 'tf.nn.tanh', 'tf.nn.sigmoid', 'tf.nn.elu', 'tf.nn.leaky_relu', 'tf.nn.softplus'])
def test_gradients_match_finite_differences(activation):
    w, x, y = _tiny(activation=activation)
    loss, g = o.member_loss_and_grads(w[0], x[0], y[0], 2)
    eps = 1e-6
    rng = np.random.default_rng(1)
    for name, arr, grad in [('W0', w[0]['W'][0], g['W'][0]), ('b1', w[0]['b'][1], g['b'][1]), ('W1', w[0]['W'][1], g['W'][1]),
                            ('W_mu', w[0]['W_mu'], g['W_mu']), ('b_var', w[0]['b_var'], g['b_var']), ('W_var', w[0]['W_var'], g['W_var'])]:
        for _ in range(5):
            idx = tuple(rng.integers(0, s) for s in arr.shape)
            old = arr[idx]
            arr[idx] = old + eps
            lp, _ = o.member_loss_and_grads(w[0], x[0], y[0], 2)
            arr[idx] = old - eps
            lm, _ = o.member_loss_and_grads(w[0], x[0], y[0], 2)
            arr[idx] = old
            fd = (lp - lm) / (2 * eps)
            assert abs(fd - grad[idx]) <= 1e-6 * max(1.0, abs(fd)), (name, idx, fd, grad[idx])


def test_nll_formula():
    # mlp_ensemble.py:64-67 on a hand case: var = 1, mu - y = 2 -> 0.5*log(2 pi) + 0.5*4
    y = np.zeros((3, 2)); mu = np.full((3, 2), 2.0); var = np.ones((3, 2))
    assert abs(o.negative_log_likelihood(y, mu, var) - (0.5 * np.log(2 * np.pi) + 2.0)) < 1e-12


def test_adam_first_step_and_clip():
    # t = 1: m = 0.1 g, v = 0.001 g^2, lr_t = lr*sqrt(0.001)/0.1 -> p -= lr * g/|g| (up to epsilon); g clipped to [-1, 1]
    p = np.array([1.0, 1.0, 1.0]); g = np.array([0.5, -3.0, 0.0])
    p2, m, v = o.adam_apply(p, g, np.zeros(3), np.zeros(3), lr=0.01, t=1)
    np.testing.assert_allclose(m, [0.05, -0.1, 0.0])                       # -3 clipped to -1
    np.testing.assert_allclose(v, [0.00025, 0.001, 0.0], rtol=1e-12)
    lr_t = 0.01 * np.sqrt(1 - 0.999) / (1 - 0.9)
    np.testing.assert_allclose(p2, [1 - lr_t * 0.05 / (np.sqrt(0.00025) + 1e-5), 1 + lr_t * 0.1 / (np.sqrt(0.001) + 1e-5), 1.0], rtol=1e-12)


def test_epoch_learning_rate_schedule():
    # mlp_ensemble.py:80-83 with training_steps 5000, train_epochs 125 (agent_factory.py:22): one decay step per fit() call
    assert o.epoch_learning_rate(0, 0.00025, 5000, 125) == np.float32(0.00025)
    assert o.epoch_learning_rate(4999, 0.00025, 5000, 125) == np.float32(0.00025)
    np.testing.assert_allclose(o.epoch_learning_rate(5000, 0.00025, 5000, 125), 0.00025 * (1 - 1 / 125), rtol=1e-6)
    assert o.epoch_learning_rate(5000 * 200, 0.00025, 5000, 125) == 0.0


def test_training_reduces_loss_on_a_learnable_problem():
    rng = np.random.default_rng(3)
    pb = o.synthetic_problem(obs_dim=4, act_dim=1, ensemble_size=2, units=16, n_layers=2, seed=5, head_scale=1.0, var_bias=0.0)
    w = o.cast_weights(pb['weights'], np.float64)
    ms, vs = o.zeros_like_weights(w), o.zeros_like_weights(w)
    A = rng.normal(0, 0.5, (5, 4))
    X = rng.normal(0, 1, (512, 5)); Y = X @ A + 0.01 * rng.normal(0, 1, (512, 4))
    first = last = None
    for t in range(1, 301):
        idx = rng.integers(0, 512, (2, 32))
        loss = o.training_step(w, ms, vs, X[idx], Y[idx], 0.01, t)
        first = loss if first is None else first
        last = loss
    assert last < first - 0.5
    assert o.validation_loss(w, X, Y) < first


def test_dropout_masks_and_masked_gradients():
    """mlp_params['dropout_rate'] (models.yaml:13): Keras Dropout keeps a unit with probability 1 - rate and scales it by
    1 / (1 - rate), in training_step only.  The masks are a pure function of (seed, step, member, layer, row, unit); the manual
    backward pass with masks is checked against finite differences of the masked forward."""
    m = o.dropout_masks(seed=7, step=3, member=1, n_layers=3, rows=64, units=128, rate=0.2)
    assert len(m) == 3 and m[0].shape == (64, 128)
    vals = np.unique(np.concatenate([x.ravel() for x in m]))
    assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 1.25) < 1e-6
    keep = np.mean([np.mean(x > 0) for x in m])
    assert abs(keep - 0.8) < 0.01
    assert not np.array_equal(m[0], m[1])                                               # layers draw different masks
    assert not np.array_equal(m[0], o.dropout_masks(7, 4, 1, 3, 64, 128, 0.2)[0])       # so do steps
    assert not np.array_equal(m[0], o.dropout_masks(7, 3, 0, 3, 64, 128, 0.2)[0])       # and members
    assert np.array_equal(m[0], o.dropout_masks(7, 3, 1, 3, 64, 128, 0.2)[0])           # and nothing else
    w, x, y = _tiny(activation='tf.nn.tanh')
    masks = o.dropout_masks(1, 0, 0, 2, x.shape[1], 8, 0.3)
    loss, g = o.member_loss_and_grads(w[0], x[0], y[0], 2, masks)
    loss0, _ = o.member_loss_and_grads(w[0], x[0], y[0], 2)
    assert abs(loss - loss0) > 1e-6
    eps = 1e-6
    rng = np.random.default_rng(2)
    for name, arr, grad in [('W0', w[0]['W'][0], g['W'][0]), ('b0', w[0]['b'][0], g['b'][0]), ('W1', w[0]['W'][1], g['W'][1]), ('W_var', w[0]['W_var'], g['W_var'])]:
        for _ in range(5):
            idx = tuple(rng.integers(0, s) for s in arr.shape)
            old = arr[idx]
            arr[idx] = old + eps
            lp, _ = o.member_loss_and_grads(w[0], x[0], y[0], 2, masks)
            arr[idx] = old - eps
            lm, _ = o.member_loss_and_grads(w[0], x[0], y[0], 2, masks)
            arr[idx] = old
            fd = (lp - lm) / (2 * eps)
            assert abs(fd - grad[idx]) <= 1e-6 * max(1.0, abs(fd)), (name, idx, fd, grad[idx])
