import sqlite3, re, sys, glob
path = sys.argv[1]; steps = int(sys.argv[2]); out = sys.argv[3]; title = sys.argv[4] if len(sys.argv) > 4 else ''
dbs = glob.glob(path + '/**/*.db', recursive=True)
db = sqlite3.connect(dbs[0]); cur = db.cursor()
rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n)
    return n[:100]
lines = ['# ' + title, '# %d steps; durations in ms; pct of total GPU kernel time' % steps,
         '%-102s %7s %10s %9s %9s %6s' % ('kernel', 'calls', 'total_ms', 'avg_ms', 'ms/step', 'pct')]
for r in rows[:40]:
    lines.append('%-102s %7d %10.2f %9.4f %9.3f %6.2f' % (short(r[0]), r[1], r[2]/1e6, r[3]/1e6, r[2]/1e6/steps, 100*r[2]/tot))
lines.append('# total GPU kernel time %.1f ms = %.2f ms/step' % (tot/1e6, tot/1e6/steps))
open(out, 'w').write('\n'.join(lines) + '\n'); print('\n'.join(lines[:32])); print(lines[-1])
