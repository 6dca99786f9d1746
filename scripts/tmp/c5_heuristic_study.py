"""CPU study: does an all-starts greedy / plateau (1,2)-swap search reach omega on the config-5 correspondence graph?"""
import os, sys, time
import numpy as np
ROOT='.'
sys.path.insert(0, ROOT)
from oracle import features as F
from oracle import oracle
C5 = np.load(os.path.join(ROOT, "tests", "golden", "config5_clouds.npz"))
A, B, vox = C5["cloud_bin_0"], C5["cloud_bin_4"], float(C5["voxel_size"])
t=time.time()
fa, _ = F.fpfh_features(A, 2 * vox, 5 * vox)
fb, _ = F.fpfh_features(B, 2 * vox, 5 * vox)
corr = F.match(fa, fb, crosscheck=True)
src=A[corr[:, 0]].astype(np.float64).T; dst=B[corr[:, 1]].astype(np.float64).T
n=src.shape[1]
_, bm = oracle.inlier_bitmap(src, dst, vox, 1.0, False)
Adj=np.unpackbits(bm.view(np.uint8), axis=1, bitorder='little')[:, :n].astype(bool)
r=oracle.max_clique(bm, n)
print("n", n, "edges", Adj.sum()//2, "omega", len(r['clique']), "unique", r['unique'], "max_core", r['max_core'], "t", round(time.time()-t,1))
deg=Adj.sum(1)
def greedy(v):
    C=[v]; P=Adj[v].copy()
    while P.any():
        idx=np.flatnonzero(P)
        d=Adj[np.ix_(idx,idx)].sum(1)
        u=idx[np.argmax(d)]   # ties: lowest index
        C.append(u); P&=Adj[u]
    return C
best=[]; sizes=[]
for v in range(n):
    if deg[v] < len(best): continue
    C=greedy(v); sizes.append(len(C))
    if len(C)>len(best): best=C
print("all-starts greedy: best", len(best), "histogram top", np.bincount(sizes)[-6:], "starts", len(sizes))
# plateau search from the 16-start style clique (take a size-(omega-1) one if any)
def plateau(C, iters=2000, seed=0):
    rng=np.random.default_rng(seed)
    C=set(C); tabu={}
    bestC=set(C)
    for it in range(iters):
        Cl=np.array(sorted(C))
        cnt=Adj[:, Cl].sum(1)
        add=[v for v in np.flatnonzero(cnt==len(Cl)) if v not in C]
        if add:
            C.add(int(add[0])); 
            if len(C)>len(bestC): bestC=set(C)
            continue
        sw=[v for v in np.flatnonzero(cnt==len(Cl)-1) if v not in C and tabu.get(int(v),-1)<it]
        if not sw: break
        v=int(sw[rng.integers(len(sw))])
        out=[c for c in Cl if not Adj[v,c]][0]
        C.remove(int(out)); C.add(v); tabu[int(out)]=it+7
    return bestC
cands=[greedy(v) for v in np.argsort(-deg)[:16]]
c0=max(cands,key=len)
print("16 top-degree starts: best", len(c0))
for s in range(3):
    print(" plateau from it ->", len(plateau(c0, seed=s)))
