import importlib, sys, time
import numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
tp = importlib.import_module("teaser-plusplus_amd")
from oracle import oracle
def popc(a):
    return np.unpackbits(a.view(np.uint8), axis=-1).sum(-1)
for (n,rho) in ((10000,0.95),(5000,0.9)):
  for seed in (20250523, 20250524, 20250530):
    pr = tp.synth_problem(seed, n, rho, 0.01)
    t=time.time()
    _, bm = oracle.inlier_bitmap(pr["src"], pr["dst"], 0.01, 1.0, False)
    deg = popc(bm).astype(np.int64)
    inl = pr["inliers"].astype(bool)
    K=int(inl.sum())
    cnt_ge = lambda j: int((deg>=j).sum())
    h0=max(k for k in range(1,n) if cnt_ge(k-1)>=k)
    print(n,rho,seed,"K",K,"h0",h0,"#out>=K-1",int((deg[~inl]>=K-1).sum()),"out deg pct 50/90/99/max",np.percentile(deg[~inl],[50,90,99,100]),"inl min",deg[inl].min(), "t",round(time.time()-t,1))
    # bits as bool matrix for R
    for klo in (K, K-50, K-100):
        R=np.flatnonzero(deg>=klo-1)
        order=R[np.argsort(-deg[R], kind='stable')]
        bits=np.unpackbits(bm[order].view(np.uint8), axis=1, bitorder='little')[:, :n][:, order].astype(np.int64)
        P=np.cumsum(bits,axis=1)
        d=deg[order][None,:]
        H=np.where(bits>0, np.minimum(P,d), 0).max(axis=1)
        h1=max([k for k in range(1,len(R)+1) if (H>=k-1).sum()>=k] or [0])
        S1=(H>=h1-1)
        print("  klo",klo,"|R|",len(R),"h1",h1,"|S1|",int(S1.sum()),"S1==inliers",set(order[S1])==set(np.flatnonzero(inl)), "H inl min/max",H[inl[order]].min(),H[inl[order]].max(),"H out max",H[~inl[order]].max() if (~inl[order]).any() else None)
