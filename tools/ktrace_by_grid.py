#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --output-format csv directory -> per (kernel name, grid size) launch count and mean / min duration.  usage: ktrace_by_grid.py DIR [substring ...]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '')
    if len(sys.argv) > 2 and not any(k in n for k in sys.argv[2:]): continue
    g = r.get('Grid_Size_X') or r.get('Grid_Size') or '?'
    d[(n[:70], g)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items()):
    print('%-72s grid %-9s calls %4d  avg %8.1f us  min %8.1f us' % (k[0], k[1], len(v), sum(v) / len(v), min(v)))
