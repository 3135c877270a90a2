"""Python host over the training half of the C ABI (cem_trainer_*): device memory and the stream come from
torch-ROCm, the training step (forward, NLL, backward, Adam) runs in libcem_mpc_gfx950.so.  The epoch / shuffle /
learning-rate loop of ``MlpEnsemble.fit`` (reference simba/models/mlp_ensemble.py:163-187) lives in
``simba/models/mlp_ensemble.py``; this class is one Keras-optimizer-plus-variables worth of state."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from .planner import _np_ptr, _ptr, flatten_weights


def unflatten_weights(blob, inputs_dim, outputs_dim, units, n_layers, ensemble_size):
    """natural blob (cem_mpc.h) -> list of per-member dicts in Keras layout."""
    out, o = [], 0
    blob = np.asarray(blob, np.float32)
    for _ in range(ensemble_size):
        Ws, bs, fi = [], [], inputs_dim
        for _ in range(n_layers):
            Ws.append(blob[o:o + fi * units].reshape(fi, units).copy()); o += fi * units
            bs.append(blob[o:o + units].copy()); o += units
            fi = units
        W_mu = blob[o:o + units * outputs_dim].reshape(units, outputs_dim).copy(); o += units * outputs_dim
        b_mu = blob[o:o + outputs_dim].copy(); o += outputs_dim
        W_var = blob[o:o + units * outputs_dim].reshape(units, outputs_dim).copy(); o += units * outputs_dim
        b_var = blob[o:o + outputs_dim].copy(); o += outputs_dim
        out.append(dict(W=Ws, b=bs, W_mu=W_mu, b_mu=b_mu, W_var=W_var, b_var=b_var))
    return out


class CemTrainer:
    """Weights + Adam moments of one MlpEnsemble on the GPU (tf.keras.optimizers.Adam(lr, clipvalue=1.0, epsilon=1e-5),
    reference mlp_ensemble.py:113-117)."""

    def __init__(self, inputs_dim, outputs_dim, units, n_layers, ensemble_size, batch_size=64, beta1=0.9, beta2=0.999,
                 epsilon=1e-5, clipvalue=1.0, device='cuda:0', activation='relu', dropout_rate=0.0, dropout_seed=0):
        import torch
        self._torch = torch
        self.lib = _capi.load()
        if not torch.cuda.is_available():
            raise RuntimeError('CemTrainer needs a ROCm GPU; there is no CPU path')
        self.dims = (inputs_dim, outputs_dim, units, n_layers, ensemble_size)
        self.batch_size, self.beta1, self.beta2 = batch_size, beta1, beta2
        c = _capi.CemTrainConfig()
        c.abi_version = _capi.CEM_ABI_VERSION
        c.inputs_dim, c.outputs_dim, c.units, c.n_layers, c.ensemble_size = inputs_dim, outputs_dim, units, n_layers, ensemble_size
        c.batch_size, c.beta1, c.beta2, c.epsilon, c.clipvalue = batch_size, beta1, beta2, epsilon, clipvalue
        from .planner import activation_code
        c.activation = activation_code(activation)
        c.dropout_rate = float(dropout_rate)                 # mlp_params['dropout_rate']: active in training_step only (mlp_ensemble.py:21,138)
        c.dropout_seed_lo, c.dropout_seed_hi = int(dropout_seed) & 0xFFFFFFFF, (int(dropout_seed) >> 32) & 0xFFFFFFFF
        self.dropout_rate, self.dropout_seed = float(dropout_rate), int(dropout_seed)
        self.ccfg = c
        self.device = torch.device(device)
        nbytes = self.lib.cem_trainer_workspace_bytes(C.byref(c))
        if nbytes == 0:
            nbytes = 256
        with torch.cuda.device(self.device):
            self.workspace = torch.zeros(nbytes + 256, dtype=torch.uint8, device=self.device)
            off = (-self.workspace.data_ptr()) % 256
            self._ws_view = self.workspace[off:off + nbytes]
            self.stream = torch.cuda.Stream(device=self.device)
            torch.cuda.synchronize(self.device)
            h = C.c_void_p()
            _capi.check(self.lib.cem_trainer_create(C.byref(c), _ptr(self._ws_view), nbytes, C.c_void_p(self.stream.cuda_stream),
                                                    C.byref(h)), 'cem_trainer_create')
        self.h = h
        self.iterations = 0                      # optimizer.iterations: persists across fit() calls

    def set_state(self, weights, m=None, v=None):
        blob = flatten_weights(weights)
        mb = flatten_weights(m) if m is not None else None
        vb = flatten_weights(v) if v is not None else None
        _capi.check(self.lib.cem_trainer_set_state(self.h, _np_ptr(blob), _np_ptr(mb), _np_ptr(vb)), 'cem_trainer_set_state')

    def get_weights(self):
        n = self.lib.cem_trainer_blob_floats(C.byref(self.ccfg))
        blob = np.empty(n, np.float32)
        _capi.check(self.lib.cem_trainer_get_state(self.h, _np_ptr(blob), None, None), 'cem_trainer_get_state')
        return unflatten_weights(blob, *self.dims)

    def get_moments(self):
        n = self.lib.cem_trainer_blob_floats(C.byref(self.ccfg))
        m, v = np.empty(n, np.float32), np.empty(n, np.float32)
        _capi.check(self.lib.cem_trainer_get_state(self.h, None, _np_ptr(m), _np_ptr(v)), 'cem_trainer_get_state')
        return unflatten_weights(m, *self.dims), unflatten_weights(v, *self.dims)

    def lr_t(self, lr):
        """Keras folds Adam's bias correction into the step size: lr * sqrt(1 - beta2^t) / (1 - beta1^t), t = iterations + 1."""
        t = self.iterations + 1
        f = np.float32
        return float(f(lr) * f(np.sqrt(1.0 - self.beta2 ** t)) / f(1.0 - self.beta1 ** t))

    def step(self, x_dev, y_dev, perm_dev, offset, bt, lr, loss_dev):
        """One MlpEnsemble.training_step (mlp_ensemble.py:134-145) on rows perm[m, offset:offset+bt] of x_dev / y_dev."""
        self.stream.wait_stream(self._torch.cuda.current_stream(self.device))
        nperm = perm_dev.shape[1] if perm_dev is not None else 0
        _capi.check(self.lib.cem_trainer_step(self.h, _ptr(x_dev), _ptr(y_dev), _ptr(perm_dev), nperm, offset, bt, self.lr_t(lr),
                                              _ptr(loss_dev)), 'cem_trainer_step')
        self.iterations += 1

    def steps(self, x_dev, y_dev, perm_dev, offsets, bts, lrs, loss_dev):
        """len(offsets) consecutive training_steps in one library call (an epoch's inner loop): step s takes rows
        perm[m, offsets[s]:offsets[s]+bts[s]] with learning rate lrs[s]; loss_dev is [n_steps, E]."""
        n = len(offsets)
        off = np.ascontiguousarray(np.asarray(offsets, np.int32))
        bt = np.ascontiguousarray(np.asarray(bts, np.int32))
        f = np.float32
        t = self.iterations + 1 + np.arange(n)
        lr_t = np.ascontiguousarray((np.asarray(lrs, f) * np.sqrt(1.0 - self.beta2 ** t).astype(f) / (1.0 - self.beta1 ** t).astype(f)).astype(f))
        self.stream.wait_stream(self._torch.cuda.current_stream(self.device))
        _capi.check(self.lib.cem_trainer_steps(self.h, _ptr(x_dev), _ptr(y_dev), _ptr(perm_dev), perm_dev.shape[1], n, _np_ptr(off),
                                               _np_ptr(bt), _np_ptr(lr_t), _ptr(loss_dev)), 'cem_trainer_steps')
        self.iterations += n

    def validation_loss(self, x_dev, y_dev):
        """MlpEnsemble.validation_step (mlp_ensemble.py:147-155)."""
        self.stream.wait_stream(self._torch.cuda.current_stream(self.device))
        out = C.c_float()
        _capi.check(self.lib.cem_trainer_eval(self.h, _ptr(x_dev), _ptr(y_dev), x_dev.shape[0], C.byref(out)), 'cem_trainer_eval')
        return float(out.value)

    def synchronize(self):
        self.stream.synchronize()

    def close(self):
        if getattr(self, 'h', None):
            self.lib.cem_trainer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
