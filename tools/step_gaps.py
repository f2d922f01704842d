#!/usr/bin/env python3
"""Busy vs idle GPU time per step from a rocprofv3 --kernel-trace database, steps delimited by a marker kernel that runs once per step (default: the stem conv).
usage: step_gaps.py <rocprof dir> [marker substring] [last N steps]"""
import glob, sqlite3, sys
path = sys.argv[1]; marker = sys.argv[2] if len(sys.argv) > 2 else 'stem_conv_fwd'; last = int(sys.argv[3]) if len(sys.argv) > 3 else 10
db = sqlite3.connect(glob.glob(path + '/**/*.db', recursive=True)[0])
rows = list(db.execute("select name, start, end from kernels order by start"))
marks = [i for i, r in enumerate(rows) if marker in r[0]]
steps = []
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a:b]
    wall = (rows[b][1] - seg[0][1]) / 1e3
    busy = sum(e - s for _, s, e in seg) / 1e3
    gaps = sorted(((seg[i + 1][1] - seg[i][2]) / 1e3, seg[i][0][:50], seg[i + 1][0][:50]) for i in range(len(seg) - 1))
    tail = (rows[b][1] - seg[-1][2]) / 1e3
    steps.append((wall, busy, len(seg), gaps[-3:], tail))
for wall, busy, n, g, tail in steps[-last:]:
    print('step: wall %8.1f us  busy %8.1f us  idle %7.1f us  kernels %4d  gap to the next step %7.1f us | largest gaps: %s' % (wall, busy, wall - busy, n, tail, '; '.join('%.1f us after %s' % (x[0], x[1]) for x in reversed(g))))
