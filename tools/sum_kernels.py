import csv,glob,sys
f=glob.glob('gpurun_out/prof_%s/**/*kernel_stats.csv'%sys.argv[1],recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total per step (30 steps): %.1f us"%(tot/30/1e3))
for r in rows[:10]:
    print('%-70s calls %4s avg %9.1f per-step %8.1f us'%(r['Name'][:70], r['Calls'], float(r['AverageNs']), float(r['TotalDurationNs'])/30/1e3))
