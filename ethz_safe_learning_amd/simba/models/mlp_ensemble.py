"""MlpEnsemble, reference simba/models/mlp_ensemble.py:91-193: E independent Gaussian MLPs, L x (Dense U + ReLU) ->
(mu Dense, softplus+1e-4 var Dense).  Holds the weights in Keras layout ([in, out], Glorot-uniform kernels, zero
biases: Keras Dense defaults of mlp_ensemble.py:13,28-29).  Inference happens in the fused HIP rollout kernel;
``fit`` (mlp_ensemble.py:163-187) keeps the reference's host loop (train/validation split, per-epoch bootstrap
shuffles, np.array_split batches, EpochLearningRateSchedule :70-88) and runs every training_step (:134-145) on the
GPU through cem_trainer_step (forward, NLL, backward, Adam with clipvalue=1 / epsilon=1e-5)."""
import logging

import numpy as np

logger = logging.getLogger('simba')


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
        from ...planner import activation_code
        self.activation = self.mlp_params.get('activation', 'tf.nn.relu')       # the reference evals this string (mlp_ensemble.py:14)
        activation_code(self.activation)                                         # raises NotImplementedError for what is not built
        self.dropout_rate = float(self.mlp_params.get('dropout_rate', 0.0))       # Dropout after every hidden layer, training_step only (mlp_ensemble.py:15,21,138)
        if not 0.0 <= self.dropout_rate < 1.0:
            raise ValueError('dropout_rate must be in [0, 1)')
        # Keras draws Dropout masks from TensorFlow's global generator; the counterpart here is numpy's (the one the reference's own
        # split and shuffles use, seeded by scripts/train.py): an unseeded model with dropout takes its mask seed from it, so two runs
        # only share a mask stream if they share np.random's state.  (No draw when the rate is 0 — the shipped value — so np.random's
        # stream is untouched there.)  The mask STREAM is parity-unpinned by construction; rate and 1 / (1 - rate) scaling are pinned.
        if seed is not None:
            self._dropout_seed = int(seed)
        else:
            self._dropout_seed = int(np.random.randint(0, 2 ** 31 - 1)) if self.dropout_rate > 0.0 else 0
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

    # ---- training (mlp_ensemble.py:70-88,134-187) ----------------------------------------------------------------
    def learning_rate_at(self, step):
        """EpochLearningRateSchedule.__call__ (:80-83) or the constant rate (:115)."""
        f = np.float32
        if not self.learning_rate_schedule:
            return f(self.learning_rate)
        epochs_so_far = f(np.floor(int(step) / int(self.training_steps)))
        return max(f(self.learning_rate) * (f(1.0) - epochs_so_far / f(self.train_epochs)), f(0.0))

    def split_train_validate(self, inputs, targets):
        indices = np.random.permutation(inputs.shape[0])                                   # :157-161
        num_val = int(inputs.shape[0] * self.validation_split)
        train_idx, val_idx = indices[num_val:], indices[:num_val]
        return inputs[train_idx, ...], targets[train_idx, ...], inputs[val_idx, ...], targets[val_idx, ...]

    def _get_trainer(self, device='cuda:0'):
        from ...trainer import CemTrainer
        if getattr(self, '_trainer', None) is None:
            self._trainer = CemTrainer(self.inputs_dim, self.outputs_dim, self.mlp_params['units'], self.mlp_params['n_layers'],
                                       self.ensemble_size, batch_size=self.batch_size, device=device, activation=self.activation,
                                       dropout_rate=self.dropout_rate, dropout_seed=self._dropout_seed)
            self._trainer_version = None
        if self._trainer_version != self.version:           # weights were replaced from outside: Adam moments restart
            self._trainer.set_state(self._weights)
            self._trainer_version = self.version
        return self._trainer

    def fit(self, inputs, targets):
        import torch
        assert inputs.shape[0] == targets.shape[0], "Inputs batch size ({}) doesn't match targets batch size ({})".format(
            inputs.shape[0], targets.shape[0])
        assert np.isfinite(inputs).all() and np.isfinite(targets).all(), "Training data is not finite."      # :166
        tr = self._get_trainer()
        train_inputs, train_targets, validate_inputs, validate_targets = self.split_train_validate(
            np.asarray(inputs, np.float32), np.asarray(targets, np.float32))
        n_train = train_inputs.shape[0]
        n_batches = int(np.ceil(n_train / self.batch_size))                                                   # :169
        dev = tr.device
        x_dev = torch.from_numpy(np.ascontiguousarray(train_inputs)).to(dev)
        y_dev = torch.from_numpy(np.ascontiguousarray(train_targets)).to(dev)
        xv_dev = torch.from_numpy(np.ascontiguousarray(validate_inputs)).to(dev) if validate_inputs.shape[0] else None
        yv_dev = torch.from_numpy(np.ascontiguousarray(validate_targets)).to(dev) if validate_inputs.shape[0] else None
        loss_dev = torch.zeros((self.training_steps, self.ensemble_size), dtype=torch.float32, device=dev)
        bounds = np.cumsum([0] + [len(a) for a in np.array_split(np.arange(n_train), n_batches)])             # np.array_split sizes, :174
        step = 0
        log_every = max(1, int(self.training_steps / 10))
        # The host runs epochs ahead of the device (a step is ~0.2 ms, nothing here synchronises): every epoch's permutation
        # tensor is read asynchronously by the steps queued on the trainer's stream, so it must not go back to torch's caching
        # allocator (which only knows the stream it was allocated on) before those steps ran.  record_stream() tells the
        # allocator about the trainer's stream; the list keeps the tensors alive until the final synchronize as well.
        perms_alive = []
        while step < self.training_steps:
            shuffles_per_mlp = np.array([np.random.permutation(n_train) for _ in range(self.ensemble_size)])  # :172-173
            perm_dev = torch.from_numpy(shuffles_per_mlp.astype(np.int32)).to(dev)
            perm_dev.record_stream(tr.stream)
            perms_alive.append(perm_dev)
            # the epoch's steps go down in runs of one library call each (cem_trainer_steps), cut where the reference logs a
            # validation loss (:181-184) and at the end of training: no Python between the steps of a run
            b = 0
            while b < n_batches and step < self.training_steps:
                until_log = log_every - step % log_every if xv_dev is not None else n_batches
                run = min(n_batches - b, self.training_steps - step, until_log)
                offs = bounds[b:b + run]
                bts = bounds[b + 1:b + run + 1] - bounds[b:b + run]
                lrs = [self.learning_rate_at(tr.iterations + i) for i in range(run)]
                tr.steps(x_dev, y_dev, perm_dev, offs, bts, lrs, loss_dev[step:step + run])
                step += run
                b += run
                if step % log_every == 0 and xv_dev is not None:                                              # :181-184
                    vl = tr.validation_loss(xv_dev, yv_dev)
                    logger.debug("Step {} | Training Loss {} | Validation Loss {}".format(
                        step, float(loss_dev[step - 1].sum().item()), vl))
        tr.synchronize()
        del perms_alive
        losses = loss_dev.sum(dim=1).cpu().numpy().astype(np.float64)
        self._weights = tr.get_weights()
        self.version += 1
        self._trainer_version = self.version
        return losses
