"""examples/capi_minimal.py: the C ABI driven by bare ctypes (the binding INTEGRATION.md section 2 describes) runs and is
deterministic, independently of this repo's planner.py wrapper."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_capi_minimal_example_runs_and_is_deterministic():
    sys.path.insert(0, os.path.join(ROOT, 'examples'))
    import capi_minimal
    a1, s1, it1 = capi_minimal.main(seed=3)
    a2, s2, it2 = capi_minimal.main(seed=3)
    assert a1.shape == (2,) and np.isfinite(a1).all() and np.all(np.abs(a1) <= 1.0 + 0.1)
    assert it1 == 4 and np.isfinite(s1)
    np.testing.assert_array_equal(a1, a2)
    assert s1 == s2 and it1 == it2
    a3, s3, _ = capi_minimal.main(seed=4)
    assert not np.array_equal(a1, a3)


def test_build_then_smoke_in_one_fresh_process():
    """`__graft_entry__.build()` dlopens the library (symbol check) before anything has imported torch; `smoke()` in the same process
    must still run: the library has to end up on the HIP runtime torch brought, not on a second copy from /opt/rocm (`_capi.load`
    imports torch first for that reason; without it `cem_planner_create` fails with hipErrorNoDevice)."""
    import subprocess
    r = subprocess.run([sys.executable, '-c', 'import __graft_entry__ as g; g.build(); g.smoke(); print("both ok")'], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'both ok' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_a_host_that_loads_the_library_before_its_hip_runtime_is_told():
    """The trap behind the test above, for a host that is not this repo's Python binding: dlopen libcem_mpc_gfx950.so FIRST (it pulls
    in /opt/rocm's libamdhip64), then bring the runtime that owns the device memory (torch's copy) — two HIP runtimes in one process.
    The first cem_planner_create says so on stderr (cem_capi.hip warn_if_two_hip_runtimes), whatever the call then returns."""
    import subprocess
    code = r'''
import ctypes as C, os, sys
root = %r
sys.path.insert(0, root)
lib_path = os.path.join(root, 'ethz_safe_learning_amd', 'lib', 'libcem_mpc_gfx950.so')
C.CDLL(lib_path)                                   # before torch: the wrong order
import torch
from ethz_safe_learning_amd import _capi
from tests import helpers as hp
pb = hp.make_problem(seed=1)
_, cfg = hp.configs(pb, N=32, H=3, P=5, E=5, k=4, I=1)
try:
    hp.make_planner(pb, cfg)
    print('create ok')
except Exception as e:
    print('create failed:', type(e).__name__)
''' % ROOT
    r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert 'create ok' in r.stdout or 'create failed' in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]
    n_runtimes = r.stderr.count('libamdhip64')
    if 'create failed' in r.stdout or n_runtimes:
        assert 'HIP runtimes are loaded in this process' in r.stderr, r.stderr[-2000:]


def test_polling_and_synchronising_hosts_get_the_same_plan():
    """cem_planner_plan on a captured graph waits for the result block by polling pinned memory; CEM_NO_POLL=1 (read once per process)
    makes it synchronise the stream instead.  Same plans either way."""
    import subprocess
    code = r'''
import sys
sys.path.insert(0, %r)
from tests import helpers as hp
pb = hp.make_problem(seed=9)
_, cfg = hp.configs(pb, N=160, H=6, P=5, E=5, k=16, I=3, noise=0.02, use_graph=True)
pl = hp.make_planner(pb, cfg)
out = [pl.plan(pb['state'], seed=2, call=c) for c in range(4)]
assert pl.graph_status() == 'graph'
print('RESULT', [(a.tolist(), s, it) for a, s, it in out])
''' % ROOT
    res = {}
    for tag, env in (('poll', {}), ('sync', {'CEM_NO_POLL': '1'})):
        r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT')][0]
    assert res['poll'] == res['sync']
