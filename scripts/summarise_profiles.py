#!/usr/bin/env python3
"""Per-kernel means of the rocprofv3 passes scripts/collect_profiles.sh wrote; output: <dir>/summary_*.csv, traffic.json."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else 'B2'
cmd = sys.argv[3] if len(sys.argv) > 3 else 'bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-split-leg --no-graph'
# 1. kernel stats: copy the rocprofv3 summary as is
for f in glob.glob(os.path.join(out, 'stats', '**', '*kernel_stats.csv'), recursive=True):
    open(os.path.join(out, 'summary_kernel_stats.csv'), 'w').write(open(f).read())
# 2. counters: mean per dispatch per kernel
rows = []
for d in sorted(glob.glob(os.path.join(out, 'pmc_*'))):
    if not os.path.isdir(d):
        continue
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r['Kernel_Name'], r['Counter_Name'])
            acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
    for (kern, ctr), (s, n) in sorted(acc.items()):
        if kern.startswith('void at::') or kern.startswith('__amd'):
            continue
        rows.append((kern, ctr, n, s / n, os.path.basename(d)))
with open(os.path.join(out, 'summary_pmc.csv'), 'w') as fh:
    fh.write('# separate rocprofv3 --kernel-trace --pmc passes of `%s` (workload %s)\n' % (cmd, workload))
    fh.write('# FETCH_SIZE / WRITE_SIZE in KB; on gfx950 FETCH_SIZE counts a 128-B request as 64 B -> x2 (MI355X_MICROARCH.md); SQ_*_CYCLES of waves are quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles\n')
    fh.write('kernel,counter,dispatches,mean_per_dispatch,pass\n')
    for r in rows:
        fh.write('"%s",%s,%d,%g,%s\n' % r)
# ONE kernel's counters: the rollout instantiation of the profiled workload (the fp32 headline path; a run that also launched the
# opt-in split kernel or another instantiation must not mix their rows in)
kern = next((k for k, c, n, m, p in rows if 'cem_rollout_' in k and 'split' not in k), None) or next((k for k, c, n, m, p in rows if 'cem_rollout_' in k), None)
roll = {c: m for k, c, n, m, p in rows if k == kern}
if roll:
    hit, miss = roll.get('TCC_HIT_sum', 0.0), roll.get('TCC_MISS_sum', 0.0)
    t = {'kernel': kern, 'workload': workload, 'fetch_size_kb': roll.get('FETCH_SIZE'), 'write_size_kb': roll.get('WRITE_SIZE'),
         'hbm_bytes_per_launch': (2 * roll.get('FETCH_SIZE', 0.0) + roll.get('WRITE_SIZE', 0.0)) * 1024,
         'correction': '2*FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts 128-B requests at 64 B)',
         'l2_hit_rate': hit / (hit + miss) if hit + miss else None,
         'sq_valu_mfma_busy_cycles_per_launch': roll.get('SQ_VALU_MFMA_BUSY_CYCLES'),
         'source': 'rocprofv3 --pmc, separate passes of `%s`' % cmd}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench                                   # the hash of the device sources this profile is of (bench.py checks it)
    t['source_sha16'] = bench.source_sha16()
    json.dump(t, open(os.path.join(out, 'traffic.json'), 'w'), indent=1)
    print(json.dumps(t))
