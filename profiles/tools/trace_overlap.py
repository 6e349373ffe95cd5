import csv, sys, glob
d=sys.argv[1]
k=[r for r in csv.DictReader(open(glob.glob(d+'/*kernel_trace.csv')[0]))]
m=[r for r in csv.DictReader(open(glob.glob(d+'/*memory_copy_trace.csv')[0]))]
ks=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:24]) for r in k)
ms=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp']),int(r.get('Bytes',r.get('Size',0)) or 0), r.get('Direction','')) for r in m)
t0=ks[0][0]
# kernel busy time (union) and gaps > 50us in the last 60% of the run
ev=sorted(ks)
busy=0; cur_s,cur_e=ev[0][0],ev[0][1]; gaps=[]
for s,e,n in ev[1:]:
    if s>cur_e:
        busy+=cur_e-cur_s
        if s-cur_e>50000: gaps.append(((cur_e-t0)/1e6,(s-cur_e)/1e3))
        cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
busy+=cur_e-cur_s
total=ev[-1][1]-t0
print('total ms',total/1e6,'kernel-busy ms',busy/1e6,'idle gaps >50us:',len(gaps), 'sum ms', sum(g for _,g in gaps)/1e3)
print('big copies (>50MB):')
big=[(s,e,b) for s,e,b,dr in ms if b>50e6]
for s,e,b in big[10:22]: print('  start %.3f ms dur %.3f ms  %.0f MB  %.1f GB/s'%((s-t0)/1e6,(e-s)/1e6,b/1e6,b/(e-s)))
print('gaps sample', gaps[5:15])
