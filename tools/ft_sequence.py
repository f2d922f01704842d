import csv, glob, sys
f = glob.glob('/tmp/prof_ft/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find the last occurrence of the SGD kernel = end of a step; print the kernels of the last full step
idx = [i for i, r in enumerate(rows) if 'sgd_multi_kernel' in r['Kernel_Name']]
a, b = idx[-2] + 1, idx[-1] + 1
tot = 0
for r in rows[a:b]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    print('%7.1f  %s' % (d, r['Kernel_Name'][:110]))
print('kernels', b - a, 'sum us', tot, 'wall us', (int(rows[b-1]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3)
