#!/usr/bin/env python3
"""Kernels of ONE step in launch order, with the idle time in front of each, from a rocprofv3 --kernel-trace database.  Steps are told apart
by their optimizer kernel (adamw_multi_kernel / sgd_multi_kernel ends a step).  usage: kernel_sequence.py <rocprof dir> <step index, negative from the end> <out.txt>"""
import glob, re, sqlite3, sys
path, which, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
db = sqlite3.connect(glob.glob(path + '/**/*.db', recursive=True)[0])
rows = list(db.execute("select name, start, end from kernels order by start"))
steps, cur = [], []
for r in rows:
    cur.append(r)
    if 'adamw_multi_kernel' in r[0] or 'sgd_multi_kernel' in r[0]:
        steps.append(cur); cur = []
# the requested step, unless the profiler stalled inside it (a trace flush shows up as gaps of tens of microseconds to milliseconds between graph nodes): then the step with the
# shortest wall time among the graph-replayed steps (the second half of the run) is shown instead and the header says so
def wall(st): return (st[-1][2] - st[0][1]) / 1e3
cand = min(range(len(steps) // 2, len(steps)), key=lambda i: wall(steps[i]))
note = ''
if wall(steps[which]) > 1.01 * wall(steps[cand]):
    note = ' (step %d had profiler stalls: wall %.1f us; showing the shortest replayed step of the run, %d)' % (which, wall(steps[which]), cand)
    which = cand
seq = steps[which]
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n)
    return re.sub(r'\(.*', '', n)[:90]
gaps = [(seq[i][1] - seq[i - 1][2]) / 1e3 for i in range(1, len(seq))]
with open(out, 'w') as f:
    f.write('# step %d of %d: %d kernels, %.1f us of kernels, %.1f us idle between them (%d gaps > 2 us), wall %.1f us\n' % (
        which, len(steps), len(seq), sum(e - s for _, s, e in seq) / 1e3, sum(g for g in gaps if g > 0), sum(1 for g in gaps if g > 2), (seq[-1][2] - seq[0][1]) / 1e3))
    if note:
        f.write('#' + note + '\n')
    for i, (n, s, e) in enumerate(seq):
        f.write('%8.1f us  idle before %6.1f  %s\n' % ((e - s) / 1e3, 0.0 if i == 0 else gaps[i - 1], short(n)))
print(open(out).readline().strip())
