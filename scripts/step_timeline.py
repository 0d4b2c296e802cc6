import csv,sys,glob
d=sys.argv[1]
ev=[]
for r in csv.DictReader(open(glob.glob(d+'/*kernel_trace.csv')[0])):
    ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:44],'K',r.get('Stream_Id','')))
for f in glob.glob(d+'/*memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Direction'][:30],'C',r.get('Stream_Id','')))
ev.sort()
idx=[i for i,e in enumerate(ev) if 'k_baq7s<' in e[2]]
i=idx[-2]
j=i
while 'k_prep_reads' not in ev[j][2]: j-=1
j-=6
t0=ev[j][0]
for k in range(j,min(len(ev),j+44)):
    s,e,n,t,st=ev[k]
    print("%9.1f %9.1f %7.1f %s %s %s"%((s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3,t,st,n))
