import numpy as np, sys
sys.path.insert(0,'/root/repo')
import bench
rng=np.random.default_rng(1)
w=bench.build_workload(24000,6.2145,1,np.random.default_rng(bench.SEED))
pos=w["pos"]%6.2145; L=6.2145; N=len(pos)
R=1.1; rc=1.0
a=(32*L**3/N)**(1/3); nc=int(round(L/a)); cw=L/nc
cx=np.minimum((pos[:,0]/cw).astype(int),nc-1); cy=np.minimum((pos[:,1]/cw).astype(int),nc-1)
serp=cx*nc+np.where(cx%2==1,nc-1-cy,cy)
zf=pos[:,2]/L; zf=np.where(serp%2==1,1-zf,zf)
order=np.lexsort((zf,serp))
P=pos[order]
nb=N//32
def morton(p):
    q=((p-p.min(0))/(np.ptp(p,axis=0)+1e-9)*3.999).astype(int)  # 4 cells per dim
    key=np.zeros(len(p),dtype=int)
    for b in range(2):
        for d in range(3):
            key|=((q[:,d]>>b)&1)<<(3*b+d)
    return np.argsort(key,kind='stable')
def blocks(morton_sort):
    B=[]
    for I in range(nb):
        p=P[I*32:(I+1)*32].copy()
        # unwrap relative to first atom
        p-=L*np.round((p-p[0])/L)
        if morton_sort: p=p[morton(p)]
        B.append(p)
    return B
def aabb_dist(lo1,hi1,lo2,hi2):
    d=np.maximum(0,np.maximum(lo1-hi2,lo2-hi1))
    return np.sqrt((d*d).sum(-1))
for ms in (False,True):
    B=blocks(ms)
    allp=np.concatenate(B); blk=np.repeat(np.arange(nb),32)
    tot_steps=0; skip=0; pairs_in=0; slots=0; skip_oct=0; tot_oct=0
    for I in rng.choice(nb,150,replace=False):
        p=B[I]; lo=p.min(0); hi=p.max(0)
        # candidate j atoms: minimum image w.r.t. block centre
        c=(lo+hi)/2
        q=allp-L*np.round((allp-c)/L)
        d=np.maximum(0,np.maximum(lo-q,q-hi)); dist=np.sqrt((d*d).sum(1))
        sel=(dist<R)&(blk!=I)
        # ownership halves this; take all and halve stats later (no effect on ratios)
        idx=np.where(sel)[0]
        qj=q[idx]
        # real pairs within rc
        dd=np.linalg.norm(p[:,None,:]-qj[None,:,:],axis=2)
        pairs_in+=(dd<rc).sum(); slots+=32*len(idx)
        # j-octets: consecutive 8 entries
        nj=len(idx)//8
        halves=[(p[:16].min(0),p[:16].max(0)),(p[16:].min(0),p[16:].max(0))]
        octs=[(p[8*k:8*k+8].min(0),p[8*k:8*k+8].max(0)) for k in range(4)]
        for k in range(nj):
            jo=qj[8*k:8*k+8]; jlo=jo.min(0); jhi=jo.max(0)
            for (hlo,hhi) in halves:
                tot_steps+=1
                if aabb_dist(hlo,hhi,jlo,jhi)>R: skip+=1
            for (olo,ohi) in octs:
                tot_oct+=1
                if aabb_dist(olo,ohi,jlo,jhi)>R: skip_oct+=1
    print("morton" if ms else "zslab","fill",pairs_in/slots,"skip frac (16x8 steps)",skip/tot_steps,"skip frac (8x8)",skip_oct/tot_oct)
