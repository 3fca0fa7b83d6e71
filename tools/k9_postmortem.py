"""Host post-mortem of a polysemanticity (K9) mismatch against scikit-learn: an fp64 numpy replay of KMeans(k, n_init=10,
random_state=123) that logs the relative margin of every decision (k-means++ searchsorted and candidate argmin, Lloyd assignments,
best-of-n_init).  A k-means++ candidate tie is STRUCTURAL, not a coincidence: two mutually nearest outliers j1, j2 (nobody else is
closer to them than the centres chosen so far) have potentials S + d(j1, j2) each — mathematically equal — and which one scikit-learn
takes hangs on the rounding of its BLAS-based distances.  `near_tie(X, k)` is what tools/fuzz_all.py consults before it calls a
mismatch a defect.    python tools/k9_postmortem.py gpurun_out/fuzz_poly_fail_seed*.npz"""
import numpy as np, sys, glob
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import oracle
from semanticlens_amd.scores import kmeans_draws
def score(centers):
    c=centers/np.maximum(np.linalg.norm(centers,axis=1,keepdims=True),1e-12)
    m=c.mean(0); n=len(c)
    return 1-((m**2).sum()-1/n)/(n-1)*n
def d2(Xc,i): return ((Xc-Xc[i])**2).sum(1)
def replay(X,k,flips=()):
    """fp64 numpy replay of sklearn KMeans(k, n_init=10, random_state=123); returns (score, min margins log)"""
    n=len(X); mean=X.mean(0); Xc=X-mean
    first,rand=kmeans_draws(n,10,123,k)
    tol=1e-4*np.mean(np.var(Xc,axis=0))
    best=None; margins=[]
    for i in range(10):
        cid=[int(first[i])]; closest=d2(Xc,cid[0]); pot=closest.sum()
        for c in range(k-1):
            rv=rand[i,c]*pot; cs=np.cumsum(closest)
            cand=np.clip(np.searchsorted(cs,rv),None,n-1)
            margins.append(("search",i,c,float(np.abs(cs[None,:]-rv[:,None]).min()/pot)))
            dc=np.stack([np.minimum(closest,d2(Xc,j)) for j in cand]); pots=dc.sum(1)
            order=np.argsort(pots,kind="stable"); b=int(order[0])
            distinct=[p for p in pots if cand[list(pots).index(p)]!=cand[b]]
            others=[pots[j] for j in range(len(cand)) if cand[j]!=cand[b]]
            if others: margins.append(("cand",i,c,float((min(others)-pots[b])/pots[b])))
            if ("cand",i,c) in flips: b=int(order[1])
            cid.append(int(cand[b])); closest=dc[b]; pot=pots[b]
        C=Xc[cid].copy()
        for it in range(300):
            dist=((Xc[:,None,:]-C[None])**2).sum(-1); lab=dist.argmin(1)
            srt=np.sort(dist,1); margins.append(("assign",i,it,float(((srt[:,1]-srt[:,0])/np.maximum(srt[:,1],1e-300)).min())))
            Cn=C.copy()
            for j in range(k):
                if (lab==j).any(): Cn[j]=Xc[lab==j].mean(0)
            # (empty clusters: sklearn relocates; ignore in this post-mortem, flagged)
            if any(not (lab==j).any() for j in range(k)): margins.append(("EMPTY",i,it,0.0))
            shift=((Cn-C)**2).sum(); C=Cn
            dist2=((Xc[:,None,:]-C[None])**2).sum(-1); lab2=dist2.argmin(1)
            if (lab2==lab).all() or shift<=tol: break
        dist=((Xc[:,None,:]-C[None])**2).sum(-1); lab=dist.argmin(1); inertia=dist.min(1).sum()
        if best is None or inertia<best[0]:
            if best is not None: margins.append(("best",i,0,float((best[0]-inertia)/best[0])))
            best=(inertia,C+mean,i)
        else:
            margins.append(("best",i,0,float((inertia-best[0])/best[0])))
    return score(best[1]),best[0],best[2],margins
def near_tie(X, k, eps=1e-12):
    """True when sklearn's procedure on X meets a decision whose relative margin is below eps (a structural tie)."""
    return any(abs(m[3]) < eps and m[0] in ("cand", "search", "assign", "best") and not (m[0] == "best" and m[3] == 0.0)
               for m in replay(np.asarray(X, np.float64), k)[3])


for f in (sorted(sys.argv[1:]) if __name__ == "__main__" else []):
    d=np.load(f); V=d['V']; nc=int(d['nc']); got=d['got']; want=d['want']
    bad=np.where(np.abs(got-want)>1e-5+1e-5*np.abs(want))[0]
    print(f.split('/')[-1], V.shape, "k",nc, "bad comps",bad, got[bad], want[bad])
    for c in bad:
        X=V[c].astype(np.float64)
        s,inertia,bi,m=replay(X,nc)
        small=sorted([x for x in m if x[0]!="EMPTY"],key=lambda x:abs(x[3]))[:4]
        print("   replay score",round(s,6),"inertia",round(inertia,5),"best init",bi," smallest margins:",small, "EMPTY" if any(x[0]=="EMPTY" for x in m) else "")
