import csv,glob,sys
f=glob.glob('gpurun_out/prof_%s/**/*kernel_stats.csv'%sys.argv[1],recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
steps=int(sys.argv[2]) if len(sys.argv)>2 else 30
print("total per step (%d steps): %.1f us"%(steps,tot/steps/1e3))
for r in rows[:int(sys.argv[3]) if len(sys.argv)>3 else 10]:
    print('%-70s calls %4s avg %9.1f per-step %8.1f us'%(r['Name'][:70], r['Calls'], float(r['AverageNs']), float(r["TotalDurationNs"])/steps/1e3))
