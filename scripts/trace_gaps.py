#!/usr/bin/env python3
"""Median gaps between consecutive kernels of the planner in a rocprofv3 --kernel-trace csv (usage: trace_gaps.py <kernel_trace.csv>):
inside a plan's graph the nodes abut; the select -> init gap is the host's turnaround between two plans."""
import collections, csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'cem_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
gaps = collections.defaultdict(list)
name = lambda r: r['Kernel_Name'].split('(')[0].split('<')[0].replace('void ', '')
for a, b in zip(rows, rows[1:]):
    gaps[(name(a), name(b))].append(int(b['Start_Timestamp']) - int(a['End_Timestamp']))
for k, v in gaps.items():
    v.sort()
    print('%-28s -> %-28s n %4d  median gap %7.2f us  min %7.2f us' % (k[0], k[1], len(v), v[len(v) // 2] / 1e3, v[0] / 1e3))
