import csv, collections, re, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
agg=collections.defaultdict(lambda:[0,0])
half=rows[len(rows)//2:]
for r in half:
    nm=re.sub(r'\(.*','',r['Kernel_Name'])[:60]; g=r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size','')
    d=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    a=agg[(nm,g)]; a[0]+=d; a[1]+=1
tot=sum(a[0] for a in agg.values())
print('total us',tot/1e3)
for k,a in sorted(agg.items(), key=lambda kv:-kv[1][0])[:int(sys.argv[2]) if len(sys.argv)>2 else 14]:
    print(f"{k[0]:62s} grid={k[1]:>9s} n={a[1]:5d} avg={a[0]/a[1]/1e3:8.1f}us tot={a[0]/tot*100:5.1f}%")
