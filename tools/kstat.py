import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in sys.argv[2:]): print('%-70s calls %s avg %.1f us'%(r['Name'][:70].replace('(anonymous namespace)::',''), r['Calls'], float(r['AverageNs'])/1e3))
