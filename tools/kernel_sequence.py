#!/usr/bin/env python3
"""Kernel names of ONE step in launch order from a rocprofv3 --kernel-trace database (to see which torch-side fills / copies sit between
the library's kernels).  usage: kernel_sequence.py <rocprof dir> <steps in the run> <out.txt>"""
import glob, re, sqlite3, sys
path, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
db = sqlite3.connect(glob.glob(path + '/**/*.db', recursive=True)[0])
rows = list(db.execute("select name, start, end from kernels order by start"))
per = len(rows) // steps
seq = rows[-per:]
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n)
    return re.sub(r'\(.*', '', n)[:90]
with open(out, 'w') as f:
    prev_end = None
    for n, s, e in seq:
        f.write('%8.1f us  gap %6.1f  %s\n' % ((e - s) / 1e3, 0.0 if prev_end is None else (s - prev_end) / 1e3, short(n)))
        prev_end = e
print('wrote', out, len(seq), 'kernels')
