#!/usr/bin/env python3
"""Diagnostic: phases of one training step of member 0 (s_memtime stamps of a -DCEM_STAMPS build, CEM_MPC_LIB=...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ethz_safe_learning_amd import synthetic
from ethz_safe_learning_amd.trainer import CemTrainer
E, D, O, U, L, n = 15, 62, 60, 128, 4, 4096
rng = np.random.default_rng(0)
tr = CemTrainer(D, O, U, L, E)
tr.set_state(synthetic.problem(O, D - O, E)['weights'])
x = torch.from_numpy(rng.standard_normal((n, D)).astype(np.float32)).cuda()
y = torch.from_numpy((0.1 * rng.standard_normal((n, O))).astype(np.float32)).cuda()
perm = torch.from_numpy(np.stack([rng.permutation(n) for _ in range(E)]).astype(np.int32)).cuda()
loss = torch.zeros(E, dtype=torch.float32, device='cuda')
for i in range(3):
    tr.step(x, y, perm, 64 * i, 64, 2.5e-4, loss)
tr.synchronize()
ws = tr._ws_view
st = ws[ws.numel() - 256:].view(torch.int64).cpu().numpy().astype(np.float64)
if os.environ.get('CEM_TRAIN_GEMM_KERNEL'):
    names = ['gather', 'forward hidden layers', 'forward heads', 'nll + grads', '(branch)', 'backward heads', 'backward layers']
    print('  '.join('%s %.1f us' % (nm, (st[i + 1] - st[i]) / 2400.0) for i, nm in enumerate(names)), ' total %.1f us' % ((st[7] - st[0]) / 2400.0))
    print('inside the %d GEMM tile passes: first slab into LDS %.1f us, k loop %.1f us, epilogue %.1f us' % (st[11], st[8] / 2400.0, st[9] / 2400.0, st[10] / 2400.0))
else:   # the tile kernel (cem_train_tile.h): workgroup 0, wave 0
    names = ['rows + inputs into LDS'] + ['forward layer %d' % l for l in range(L)] + ['heads + NLL + its gradients', 'dh_L + head dW'] + \
            ['backward layer %d (dh + dW)' % l for l in range(L - 1, -1, -1)]
    print('\n'.join('%-32s %.2f us' % (nm, (st[i + 1] - st[i]) / 2400.0) for i, nm in enumerate(names)))
    print('total %.1f us' % ((st[len(names)] - st[0]) / 2400.0))
