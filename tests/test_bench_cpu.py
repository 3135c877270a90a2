"""bench.py's host logic that needs no GPU: the guard that keeps the headline when a multi-rank extra blocks in a collective."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, time
sys.path.insert(0, %r)
import bench
out = {'metric': 'planning steps/sec (CEM-MPC, N=2000 K=5 H=30)', 'value': 123.0, 'n_gpus': 2}
with bench.HeadlineGuard(out, rank=int(sys.argv[1]), seconds=0.3, exit_code=int(sys.argv[2])):
    time.sleep(30)          # an extra blocked in a collective
print('not reached')
'''


def _run(rank, code):
    return subprocess.run([sys.executable, '-c', CHILD % ROOT, str(rank), str(code)], capture_output=True, text=True, timeout=60)


def test_headline_guard_keeps_the_headline_and_says_so():
    """A blocked extra: rank 0 prints the headline with a TOP-LEVEL `extras_timed_out: true` and error entries for the extras, every
    rank writes a stderr line, the exit status is the configured one (bench.py --extras-timeout-status; 0 by default so that a
    launcher which discards failed runs keeps the headline) — never a silent, clean-looking end."""
    r = _run(0, 0)
    assert r.returncode == 0 and 'not reached' not in r.stdout
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['value'] == 123.0 and line['extras_timed_out'] is True
    assert all('did not finish' in line[k]['error'] for k in ('b2_strong', 'b5', 'b5_split_bf16x3'))
    assert 'extras did not finish within' in r.stderr and 'extras_timed_out' in r.stderr
    r = _run(0, 3)
    assert r.returncode == 3 and json.loads(r.stdout.strip().splitlines()[-1])['extras_timed_out'] is True
    r = _run(1, 3)                      # the other ranks: no line, the same stderr notice, the same status
    assert r.returncode == 3 and r.stdout.strip() == '' and 'rank 1' in r.stderr


CHILD_RUN = r'''
import sys, time
sys.path.insert(0, %r)
import bench
g = bench.RunGuard(rank=int(sys.argv[1]), gpus=8, seconds=0.3).start()
g.stage = 'headline warm-up'
time.sleep(30)              # a collective that never completes
print('not reached')
'''


def test_run_guard_says_that_the_headline_never_finished():
    """A multi-rank headline that blocks: rank 0 prints a line with `value` null, an error and how far the run got; every rank leaves
    with a non-zero status and a stderr line — instead of the launcher's timeout and no line at all."""
    r = subprocess.run([sys.executable, '-c', CHILD_RUN % ROOT, '0'], capture_output=True, text=True, timeout=60)
    assert r.returncode == 4 and 'not reached' not in r.stdout
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['value'] is None and line['headline_timed_out'] is True and 'headline warm-up' in line['error'] and line['n_gpus'] == 8
    assert 'did not finish' in r.stderr
    r = subprocess.run([sys.executable, '-c', CHILD_RUN % ROOT, '3'], capture_output=True, text=True, timeout=60)
    assert r.returncode == 4 and r.stdout.strip() == '' and 'rank 3' in r.stderr
