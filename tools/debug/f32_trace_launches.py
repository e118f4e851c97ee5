import numpy as np, sys
f=sys.argv[1]
rec=np.load(f)['rec']
T=rec[:,2:8].astype(np.float64)/100.0
T-=T[:,0].min()
bx=(rec[:,0]>>32).astype(np.int64); by=rec[:,13].astype(np.int64)
kid=(rec[:,0]&0xffffffff).astype(np.int64)
M=(rec[:,8]&0xffffffff).astype(np.int64);K=(rec[:,8]>>32).astype(np.int64);N=(rec[:,9]&0xffffffff).astype(np.int64);grid=(rec[:,9]>>32).astype(np.int64)
trips=(rec[:,10]&0xffffffff).astype(np.int64)
bm=kid&0xff;bn=(kid>>8)&0xff;mode=(kid>>16)&0xf
starts=np.where((bx==0)&(by==0))[0]
ends=list(starts[1:])+[len(rec)]
L=[]
for s,e in zip(starts,ends):
    t=T[s:e]
    L.append(dict(s=s,e=e,M=M[s],K=K[s],N=N[s],bm=bm[s],bn=bn[s],mode=mode[s],wgs=e-s,t0=t[:,0].min(),t0l=t[:,0].max(),t1=t[:,5].max(),t1f=t[:,5].min(),
      pro=(t[:,1]-t[:,0]).mean(),first=(t[:,2]-t[:,1]).mean(),loop=(t[:,3]-t[:,2]).mean(),epi=(t[:,4]-t[:,3]).mean(),drain=(t[:,5]-t[:,4]).mean(),trips=trips[s:e].mean(),
      life=(t[:,5]-t[:,0]).sum()))
L.sort(key=lambda d:d['t0'])
print("M K N tile mode wgs | start | span | dispatch ramp (first..last start) | end ramp (first..last end) | gap to prev launch end | pro first loop epi | ns/trip | TF/s | wg-slots avg")
prev_end=None; tot_gap=0; 
for d in L:
    fl=2.0*d['M']*d['K']*d['N']
    span=d['t1']-d['t0']
    gap=(d['t0']-prev_end) if prev_end is not None else 0
    print(f"{d['M']:5d} {d['K']:5d} {d['N']:6d} {d['bm']}x{d['bn']} m{d['mode']} {d['wgs']:5d} | {d['t0']:8.1f} | {span:6.1f} | {d['t0l']-d['t0']:6.1f} | {d['t1']-d['t1f']:6.1f} | {gap:6.1f} | {d['pro']:5.2f} {d['first']:5.2f} {d['loop']:6.2f} {d['epi']:5.2f} | {d['loop']/max(d['trips'],1)*1e3:5.0f} | {fl/span/1e6:6.1f} | {d['life']/span:6.1f}")
    if prev_end is not None: tot_gap+=max(gap,0)
    prev_end=max(prev_end or 0,d['t1'])
print("total positive gaps",tot_gap, "span", max(d['t1'] for d in L)-min(d['t0'] for d in L), "sum of spans", sum(d['t1']-d['t0'] for d in L))
print("---- per-launch CU distribution")
hw=(rec[:,1]&0xffffffff).astype(np.int64); xcc=(rec[:,1]>>32).astype(np.int64)&0xf
cu=((hw>>8)&0xf)|(((hw>>12)&1)<<4)|(((hw>>13)&7)<<5)|(xcc<<8)
for d in L[12:24]:
    s,e=d['s'],d['e']
    c=cu[s:e]; u,cnt=np.unique(c,return_counts=True)
    # max concurrent per CU: sweep
    t=T[s:e]
    mx=[]
    for cc in u[:256]:
        m=c==cc
        ev=sorted([(a,1) for a in t[m,0]]+[(b,-1) for b in t[m,5]])
        cur=0;best=0
        for _,dl in ev:
            cur+=dl;best=max(best,cur)
        mx.append(best)
    print(d['M'],d['K'],d['N'],'wgs',e-s,'CUs used',len(u),'wgs/CU hist',np.bincount(cnt)[:14],'max concurrent hist',np.bincount(mx)[:10], 'xcc hist', np.bincount(xcc[s:e]))
print("---- inside one launch: s1 c2 (128,1152,25088)")
d=[x for x in L if x['M']==128 and x['K']==1152][1]
s,e=d['s'],d['e']
t=T[s:e]-T[s:e,0].min(); tr=trips[s:e]; c=cu[s:e]
lp=t[:,3]-t[:,2]
for k in np.unique(tr):
    m=tr==k
    print('trips',k,'n',m.sum(),'loop us min/med/max',lp[m].min(),np.median(lp[m]),lp[m].max(),'end min/med/max',t[m,5].min(),np.median(t[m,5]),t[m,5].max(), 'start max', t[m,0].max())
u,cnt=np.unique(c,return_counts=True)
four=set(u[cnt==4])
m4=np.array([x in four for x in c])
for nm,mm in (('CUs with 4',m4),('CUs with 3',~m4)):
    w=mm&(tr==72)
    print(nm,'whole tiles',w.sum(),'loop med',np.median(lp[w]),'end med',np.median(t[w,5]),'end max',t[w,5].max(), 'producers there', (mm&(tr!=72)).sum())
# time series: number of WGs in k-loop over time, and MFMA-equivalent rate
grid_t=np.arange(0,80,2.0)
act=[((t[:,2]<=x)&(t[:,3]>x)).sum() for x in grid_t]
print('WGs in k-loop at t=',list(zip(grid_t.astype(int),act)))
# per-WG per-trip time vs concurrent count on its CU
print("---- per-launch in-loop efficiency: sum(trips*512cyc) per CU / union of k-loop intervals on that CU")
def union_len(iv):
    iv = iv[np.argsort(iv[:, 0])]
    tot, cs, ce = 0.0, iv[0, 0], iv[0, 1]
    for s_, e_ in iv[1:]:
        if s_ > ce:
            tot += ce - cs; cs, ce = s_, e_
        else:
            ce = max(ce, e_)
    return tot + ce - cs
tot_need=0; tot_union=0; tot_span=0
for d in L:
    s,e=d['s'],d['e']
    t=T[s:e]; c=cu[s:e]; tr=trips[s:e]
    need=0; un=0
    for cc in np.unique(c):
        m=c==cc
        need+=tr[m].sum()*512/2400.0  # us of matrix-pipe time (4 SIMDs in parallel: per-SIMD 512 cycles per trip)
        un+=union_len(np.stack([t[m,2],t[m,3]],1))
    span=(d['t1']-d['t0'])*256
    tot_need+=need; tot_union+=un; tot_span+=span
    print(f"{d['M']:5d} {d['K']:5d} {d['N']:6d} m{d['mode']} wgs {d['wgs']:5d} in-loop eff {need/un*100:5.1f}%  loop coverage of launch span {un/span*100:5.1f}%  => MFMA busy over span {need/span*100:5.1f}%")
print(f"TOTAL in-loop eff {tot_need/tot_union*100:.1f}% coverage {tot_union/tot_span*100:.1f}% busy {tot_need/tot_span*100:.1f}%")
