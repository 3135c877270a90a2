#!/usr/bin/env python3
"""Same-box A/B of library builds over the BASELINE configs (scripts/sweep_configs.py per build, builds alternating):
rollout kernel ms per launch and plans/s.  MI355X boxes differ by a few % in wall time, so builds are only comparable inside
one gpurun call.  usage: ab_sweep.py a.so b.so ... [--rounds R] [--only B1,B2] [--chunks C]"""
import json, os, subprocess, sys
libs = [a for a in sys.argv[1:] if a.endswith('.so')]
def opt(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default
rounds, only, chunks = int(opt('--rounds', '2')), opt('--only', ''), opt('--chunks', '0')
here = os.path.dirname(os.path.abspath(__file__))
res = {}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, CEM_MPC_LIB=os.path.abspath(l), CEM_SWEEP_ONLY=only, CEM_SWEEP_CHUNKS=chunks)
        out = subprocess.run([sys.executable, os.path.join(here, 'sweep_configs.py')], env=env, capture_output=True, text=True)
        if out.returncode != 0:
            print(l, 'FAILED', out.stderr[-2000:]); continue
        for line in out.stdout.strip().splitlines():
            d = json.loads(line)
            res.setdefault(d['config'], {}).setdefault(l, []).append(d)
for cfg, by in res.items():
    for l in libs:
        v = by.get(l, [])
        if not v: continue
        print('%-22s %-28s rc %d wg %5d  rollout ms %s  frac %s  plans/s %s' % (
            cfg, os.path.basename(l), v[0]['chunks_per_tile'], v[0]['workgroups'], ' '.join('%.4f' % x['rollout_ms'] for x in v),
            ' '.join('%.3f' % x['frac_of_157_3'] for x in v), ' '.join('%.1f' % x['plans_per_s'] for x in v)), flush=True)
