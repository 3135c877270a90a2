#!/usr/bin/env python3
"""Diagnostic: device time of one training step / one validation pass of the ensemble trainer (HIP events around batches
of launches), E members x batch 64.  usage: python scripts/time_train_kernel.py [E]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ethz_safe_learning_amd.trainer import CemTrainer
from ethz_safe_learning_amd import synthetic

E = int(sys.argv[1]) if len(sys.argv) > 1 else 15
D, O, U, L, n = 62, 60, 128, 4, 4096
rng = np.random.default_rng(0)
tr = CemTrainer(D, O, U, L, E)
w = synthetic.problem(O, D - O, E)['weights']
tr.set_state(w)
x = torch.from_numpy(rng.standard_normal((n, D)).astype(np.float32)).cuda()
y = torch.from_numpy((0.1 * rng.standard_normal((n, O))).astype(np.float32)).cuda()
perm = torch.from_numpy(np.stack([rng.permutation(n) for _ in range(E)]).astype(np.int32)).cuda()
loss = torch.zeros((400, E), dtype=torch.float32, device='cuda')
def timed(fn, reps):
    fn(0); tr.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(tr.stream):
        e0.record()
        for i in range(reps):
            fn(i)
        e1.record()
    tr.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
t_step = timed(lambda i: tr.step(x, y, perm, (i * 64) % (n - 64), 64, 2.5e-4, loss[i % 400]), 200)
t_eval = timed(lambda i: tr.validation_loss(x[:64], y[:64]), 50)
print('E=%d: train step %.1f us, validation pass (forward only, 64 rows) %.1f us' % (E, t_step, t_eval))
