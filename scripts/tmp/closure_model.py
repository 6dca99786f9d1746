"""numpy model of the degree closure (kernels_heuristic.hip) -- debugging aid"""
import importlib, sys
import numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
tp = importlib.import_module("teaser-plusplus_amd")
from oracle import oracle
CAP=2048
def popc(a): return np.unpackbits(a.view(np.uint8), axis=-1).sum(-1)
def model(bm, n, verbose=False):
    deg = popc(bm).astype(np.int64)
    cge = lambda j: int((deg>=j).sum())
    h0 = max([k for k in range(1,n+1) if cge(k-1)>=k] or [0]); h0=min(h0,CAP)
    tmin=(h0-1)-(h0-1)//8
    ts=[j for j in range(tmin, CAP) if cge(j)<=CAP]
    if not ts: return ("decline: no t",)
    t=ts[0]; m=cge(t)
    if h0-1<t or m<2: return ("decline: h0-1<t", h0, t, m)
    R=np.flatnonzero(deg>=t)
    order=R[np.lexsort((R, -deg[R]))]
    bits=np.unpackbits(bm[order].view(np.uint8), axis=1, bitorder='little')[:, :n][:, order].astype(np.int64)
    P=np.cumsum(bits,axis=1); d=deg[order][None,:]
    H=np.where(bits>0, np.minimum(P,d), 0).max(axis=1)
    kmax=CAP
    for attempt in range(8):
        ks=[k for k in range(2,min(kmax,m)+1) if (H>=k-1).sum()>=k]
        if not ks: return ("decline: no feasible k", h0,t,m)
        k=ks[-1]
        if k-1<t: return ("decline: k-1<t", h0,t,m,k)
        S=H>=k-1
        while True:
            c=(bits[:,S].sum(axis=1))
            S2=S&(c>=k-1)
            if S2.sum()==S.sum() or S2.sum()<k: S=S2; break
            S=S2
        if S.sum()>k: return ("decline: core > k", h0,t,m,k,int(S.sum()))
        if S.sum()<k: kmax=k-1; continue
        return ("closed", h0,t,m,k, sorted(order[S].tolist()))
    return ("decline: attempts",)
if __name__=="__main__":
    n,rho=int(sys.argv[1]),float(sys.argv[2])
    for seed in map(int, sys.argv[3:]):
        pr = tp.synth_problem(seed, n, rho, 0.01)
        _, bm = oracle.inlier_bitmap(pr["src"], pr["dst"], 0.01, 1.0, False)
        r=model(bm,n)
        print(seed, r[:5], "K", int(pr["inliers"].sum()))
