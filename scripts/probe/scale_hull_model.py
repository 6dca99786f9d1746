"""Sizing of a sort-free scale stage BEFORE building it: which fraction of the 2M TRIM endpoints would an
exact sweep still have to sort if value bins (exact prefix sums at the bin boundaries) pruned every bin whose LOWER
bound on the cost -- R_total - (stabbing weight at the bin start + the openers inside), optionally + the sum of squares of
the members that cannot leave inside the bin -- exceeds one achieved cost?  CPU / numpy model on the bench's scale workload
(tests/golden/scale_golden.json entry 0) at a reduced N.  usage: scale_hull_model.py [n]   (needs the built library for
synth_problem only; results: profiles/r6b/scale_hull_model.txt)"""
import importlib, sys, json, numpy as np
sys.path.insert(0,'/root/repo')
tp = importlib.import_module("teaser-plusplus_amd")
g = json.load(open('/root/repo/tests/golden/scale_golden.json'))[0]
n = int(sys.argv[1]) if len(sys.argv)>1 else 3000
rho, k, nb = float(g["outlier_ratio"]), float(g["dst_scale"]), float(g["noise_bound"])
pr = tp.synth_problem(int(g["seed"]), n, rho, 0.01)
src, dst = pr["src"], pr["dst"]*k
iu = np.triu_indices(n,1)
a = np.linalg.norm(src[:,iu[1]]-src[:,iu[0]],axis=0); b = np.linalg.norm(dst[:,iu[1]]-dst[:,iu[0]],axis=0)
beta = 2*nb
x = b/a; r = beta/a
M = len(x); print("n",n,"M",M,"beta",beta, "x med",np.median(x), "r med",np.median(r), "r max", r.max(), "x max", x.max())
lo = x-r; hi = x+r
# exact sweep (numpy): sort endpoints
ev = np.concatenate([lo,hi]); sg = np.concatenate([np.ones(M),-np.ones(M)]); idx = np.concatenate([np.arange(M),np.arange(M)])
o = np.argsort(ev,kind='stable'); ev=ev[o]; sg=sg[o]; idx=idx[o]
w = 1/r**2
cs = lambda v: np.cumsum(sg*v[idx])
card=cs(np.ones(M)); sw=cs(w); swx=cs(w*x); sr=cs(r); sx=cs(x); sxx=cs(x*x)
xh = swx/sw
cost = card*xh*xh + sxx - 2*sx*xh + (r.sum()-sr)
cost[~np.isfinite(cost)] = np.inf
am = np.argmin(cost); print("argmin pos frac", am/(2*M), "value", ev[am], "xhat", xh[am], "cost", cost[am], "Rtotal", r.sum(), "S top", sr.max(), "card at min", card[am])
# bins
for B in (1024, 4096, 16384):
    X=8.0
    bo = np.clip((lo/X*B).astype(np.int64),0,B-1); bc = np.clip((hi/X*B).astype(np.int64),0,B-1)
    orr = np.bincount(bo,weights=r,minlength=B); crr = np.bincount(bc,weights=r,minlength=B)
    Sstart = np.concatenate([[0],np.cumsum(orr-crr)[:-1]])
    Shat = Sstart + orr
    # UB: cost at boundary with max Sstart: exact cost from sweep at last event < boundary
    b0 = np.argmax(Sstart); t0 = b0*X/B
    p0 = np.searchsorted(ev,t0,side='left')-1
    UB = cost[p0] if p0>=0 else np.inf
    Smin = r.sum()-UB
    cand = Shat >= Smin
    first, last = np.flatnonzero(cand)[[0,-1]]
    tlo, thi = first*X/B, (last+1)*X/B
    inhull = ((ev>=tlo)&(ev<thi)).sum()
    print("B",B,"UB",UB,"true min",cost[am],"cand bins",cand.sum(),"hull",tlo,thi,"endpoints in hull frac",inhull/(2*M), "argmin inside", tlo<=ev[am]<thi)
    # with SS lower bound at bin level (exact SS of core set): SS_core = SS over I_start \ D
    oc = np.bincount(bo,minlength=B); cc=np.bincount(bc,minlength=B)
    ox = np.bincount(bo,weights=x,minlength=B); cx=np.bincount(bc,weights=x,minlength=B)
    oxx= np.bincount(bo,weights=x*x,minlength=B); cxx=np.bincount(bc,weights=x*x,minlength=B)
    pre = lambda o_,c_: np.concatenate([[0],np.cumsum(o_-c_)[:-1]])
    n0=pre(oc,cc); s1=pre(ox,cx); s2=pre(oxx,cxx)
    nc=n0-cc; s1c=s1-cx; s2c=s2-cxx
    with np.errstate(all='ignore'):
        SScore = np.where(nc>0, s2c - s1c*s1c/np.maximum(nc,1), 0.0)
    SScore=np.maximum(SScore,0)
    LB = r.sum() - Shat + SScore
    cand2 = LB <= UB
    f2,l2 = np.flatnonzero(cand2)[[0,-1]]
    t2lo,t2hi=f2*X/B,(l2+1)*X/B
    print("   with SS core bound: cand bins",cand2.sum(),"hull frac",((ev>=t2lo)&(ev<t2hi)).sum()/(2*M),"argmin inside", t2lo<=ev[am]<t2hi)
print("---- nonlinear bins: linear on [0,4) with B-256 bins, then geometric up to 1e4 in 255 bins, last catch-all")
def binmap(v,B):
    Bl=B-256
    lin = np.floor(np.clip(v,0,None)/4.0*Bl)
    tail = Bl + np.floor(np.log(np.maximum(v,4.0)/4.0)/np.log(1e4/4.0)*255)
    return np.clip(np.where(v<4.0,lin,tail),0,B-1).astype(np.int64)
for B in (1024,2048,4096):
    bo=binmap(lo,B); bc=binmap(hi,B)
    orr = np.bincount(bo,weights=r,minlength=B); crr = np.bincount(bc,weights=r,minlength=B)
    Sstart = np.concatenate([[0],np.cumsum(orr-crr)[:-1]]); Shat=Sstart+orr
    b0=np.argmax(Sstart)
    # boundary value of bin b0: smallest v with binmap(v)=b0 -> for linear part b0*4/Bl
    Bl=B-256; t0=b0*4.0/Bl
    p0=np.searchsorted(ev,t0,side='left')-1; UB=cost[p0]; Smin=r.sum()-UB
    oc=np.bincount(bo,minlength=B); cc=np.bincount(bc,minlength=B)
    ox=np.bincount(bo,weights=x,minlength=B); cx=np.bincount(bc,weights=x,minlength=B)
    oxx=np.bincount(bo,weights=x*x,minlength=B); cxx=np.bincount(bc,weights=x*x,minlength=B)
    pre=lambda o_,c_: np.concatenate([[0],np.cumsum(o_-c_)[:-1]])
    n0=pre(oc,cc); s1=pre(ox,cx); s2=pre(oxx,cxx); nc=n0-cc; s1c=s1-cx; s2c=s2-cxx
    SScore=np.maximum(np.where(nc>0, s2c - s1c*s1c/np.maximum(nc,1), 0.0),0)
    for name,cand in (("r only",Shat>=Smin),("with SS",(r.sum()-Shat+SScore)<=UB)):
        ids=np.flatnonzero(cand); first,last=ids[0],ids[-1]
        evb=binmap(ev,B)
        inc=cand[evb].sum(); inh=((evb>=first)&(evb<=last)).sum()
        print("B",B,name,"cand bins",cand.sum(),"range",first,last,"endpoints in cand bins",inc/(2*M),"in hull",inh/(2*M),"argmin bin in", cand[binmap(ev[am:am+1],B)[0]])
