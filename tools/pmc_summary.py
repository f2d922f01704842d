import sqlite3, sys, glob, re
from collections import defaultdict
path = sys.argv[1]
db = sqlite3.connect(glob.glob(path + '/**/*.db', recursive=True)[0]); cur = db.cursor()
rows = list(cur.execute("select name, dispatch_id, duration, counter_name, counter_value from pmc_events"))
agg = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
for name, did, d, cn, cv in rows:
    n = re.sub(r'\(anonymous namespace\)::', '', name); n = re.sub(r'^void ', '', n)[:70]
    agg[n][cn].append(cv); dur[n].append(d)
for n in agg:
    if not any(k in n for k in (sys.argv[2:] or ['conv', 'wgrad'])): continue
    print('==', n)
    for cn, v in sorted(agg[n].items()):
        print('   %-28s mean %14.1f  (n=%d)' % (cn, sum(v) / len(v), len(v)))
    c = {k: sum(v) / len(v) for k, v in agg[n].items()}
    if 'SQ_WAVE_CYCLES' in c:
        wc = c['SQ_WAVE_CYCLES']
        for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY'):
            if k in c: print('   %s / WAVE_CYCLES = %.3f' % (k, c[k] / wc))
    if 'SQ_LDS_BANK_CONFLICT' in c and c.get('SQ_LDS_IDX_ACTIVE'):
        print('   LDS bank conflict / idx active = %.3f' % (c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'SQ_BUSY_CYCLES' in c:
        print('   MFMA_BUSY / BUSY_CYCLES = %.4f' % (c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES']))
