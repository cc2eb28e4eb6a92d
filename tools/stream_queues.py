import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
ev=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:50],r['Stream_Id'],r['Queue_Id']) for r in rows)
st=[i for i,e in enumerate(ev) if 'vox_bucket' in e[2]]
a,b=st[-3],st[-2]; step=ev[a:b]
print('period', (ev[b][0]-ev[a][0])/1e3, 'sum', sum(e[1]-e[0] for e in step)/1e3, sorted(set((e[3],e[4]) for e in step)))
