import csv,collections,sys
f=sys.argv[1]
d=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'wgrad_wino16' in n or 'wgrad_m16' in n:
        k=(n.split('(')[0][5:45], r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Grid_Size_Y'))
        d.setdefault(k,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items():
    if len(v)>5:
        v=sorted(v); print(k,len(v),'median %.1f min %.1f'%(v[len(v)//2],v[0]))
