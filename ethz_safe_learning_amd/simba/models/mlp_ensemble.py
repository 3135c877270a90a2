"""MlpEnsemble (inference half), reference simba/models/mlp_ensemble.py:91-132,189-193: E independent Gaussian
MLPs, L x (Dense U + ReLU) -> (mu Dense, softplus+1e-4 var Dense).  Holds the weights in Keras layout ([in, out],
Glorot-uniform kernels, zero biases: Keras Dense defaults of mlp_ensemble.py:13,28-29); evaluation happens in the
fused HIP kernel.  ``fit`` (mlp_ensemble.py:163-187) is SURVEY 8f row 1 ("next") and not implemented yet."""
import numpy as np


class MlpEnsemble(object):
    def __init__(self, inputs_dim, outputs_dim, ensemble_size, batch_size=64, validation_split=0.2, learning_rate=0.00025,
                 learning_rate_schedule=True, training_steps=5000, mlp_params=None, train_epochs=1, seed=None):
        self.inputs_dim = inputs_dim
        self.outputs_dim = outputs_dim
        self.ensemble_size = ensemble_size
        self.batch_size = batch_size
        self.validation_split = validation_split
        self.learning_rate = learning_rate
        self.learning_rate_schedule = learning_rate_schedule
        self.training_steps = training_steps
        self.train_epochs = train_epochs
        self.mlp_params = dict(mlp_params or dict(n_layers=4, units=128, activation='tf.nn.relu', dropout_rate=0.0))
        act = self.mlp_params.get('activation', 'tf.nn.relu')
        if act not in ('tf.nn.relu', 'relu'):
            raise NotImplementedError('only the relu activation of config/models.yaml:12 is built into the kernel')
        if float(self.mlp_params.get('dropout_rate', 0.0)) != 0.0:
            raise NotImplementedError('dropout is identity at inference (mlp_ensemble.py:127); training is not built yet')
        rng = np.random.default_rng(seed)
        self._weights = [self._init_member(rng) for _ in range(ensemble_size)]
        self.version = 0

    def _init_member(self, rng):
        U, L = self.mlp_params['units'], self.mlp_params['n_layers']

        def glorot(fi, fo):
            lim = np.sqrt(6.0 / (fi + fo))
            return rng.uniform(-lim, lim, size=(fi, fo)).astype(np.float32)
        Ws, bs, fi = [], [], self.inputs_dim
        for _ in range(L):
            Ws.append(glorot(fi, U)); bs.append(np.zeros((U,), np.float32)); fi = U
        return dict(W=Ws, b=bs, W_mu=glorot(U, self.outputs_dim), b_mu=np.zeros((self.outputs_dim,), np.float32),
                    W_var=glorot(U, self.outputs_dim), b_var=np.zeros((self.outputs_dim,), np.float32))

    def build(self):
        pass

    def get_weights(self):
        return self._weights

    def set_weights(self, weights):
        """Replace all members' weights (list of dict(W, b, W_mu, b_mu, W_var, b_var), Keras [in, out] layout)."""
        assert len(weights) == self.ensemble_size
        self._weights = weights
        self.version += 1

    def fit(self, inputs, targets):
        raise NotImplementedError('MlpEnsemble.fit (mlp_ensemble.py:163-187) is the next hot-path row (SURVEY 8f-1); '
                                  'load trained weights with set_weights()')
