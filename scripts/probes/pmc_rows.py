import csv,sys,glob,collections
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
d=collections.OrderedDict()
for r in rows:
    if 'trace' not in r['Kernel_Name']: continue
    k=r['Dispatch_Id']
    d.setdefault(k,{})[r['Counter_Name']]=float(r['Counter_Value'])
for k,v in d.items():
    print(k,' '.join(f"{n}={x:.4g}" for n,x in sorted(v.items())))
