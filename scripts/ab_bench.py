#!/usr/bin/env python3
"""Same-box A/B of library builds: MI355X devices differ by up to ~12 % in wall time, so variants are only comparable
inside one gpurun call.  usage: ab_bench.py lib1.so lib2.so ... [--rounds R] [--chunks C]"""
import json, os, subprocess, sys
libs = [a for a in sys.argv[1:] if a.endswith('.so')]
rounds = int(sys.argv[sys.argv.index('--rounds') + 1]) if '--rounds' in sys.argv else 2
chunks = sys.argv[sys.argv.index('--chunks') + 1] if '--chunks' in sys.argv else '0'
steps = sys.argv[sys.argv.index('--steps') + 1] if '--steps' in sys.argv else '100'
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, CEM_MPC_LIB=os.path.abspath(l))
        out = subprocess.run([sys.executable, 'bench.py', '--steps', steps, '--warmup', '15', '--no-cpu-baseline', '--no-split-leg', '--no-configs', '--chunks', chunks],
                             env=env, capture_output=True, text=True).stdout.strip().splitlines()
        d = json.loads(out[-1])
        res[l].append((d['roofline']['avg_launch_ms'], d['value'], d['ms_per_step_median']))
for l in libs:
    print('%-40s rollout ms %s   plans/s %s   median plan ms %s' % (os.path.basename(l), ' '.join('%.4f' % a for a, _, _ in res[l]), ' '.join('%.1f' % b for _, b, _ in res[l]),
                                                                       ' '.join('%.4f' % c for _, _, c in res[l])), flush=True)
