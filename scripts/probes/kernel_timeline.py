import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
n=len(rows)
for r in rows[max(0,n-40):]:
    s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    print(f"{s:10.1f} {e:10.1f} {e-s:8.1f} q{r.get('Queue_Id','?')} {r['Kernel_Name'][:40]} grid {r.get('Grid_Size_X', r.get('Grid_Size','?'))}")
