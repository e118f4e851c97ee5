import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'stats_reset' in r['Kernel_Name']]
last=rows[idx[-1]:]
tot=0; out=[]
for r in last:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    tot+=d
    nm=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:34]
    out.append(f"{d:6.1f} {nm:34s} {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
print('\n'.join(out)); print('sum',tot)
