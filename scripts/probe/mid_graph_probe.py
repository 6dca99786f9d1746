import importlib, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
tp = importlib.import_module("teaser-plusplus_amd")
from oracle import oracle
s = tp.RobustRegistrationSolver(tp.RobustRegistrationSolver.Params())
rng = np.random.default_rng(16)
for n, p in ((600, 0.12), (760, 0.10), (900, 0.08), (1020, 0.08), (1080, 0.07)):
    A = np.triu(rng.uniform(size=(n, n)) < p, 1)
    members = np.sort(rng.choice(n, size=9, replace=False))
    A[np.ix_(members, members)] |= np.triu(np.ones((9, 9), dtype=bool), 1)
    bm = oracle.bitmap_from_edges(n, np.argwhere(A))
    c, er = s.maxClique(bm, n)
    print("n", n, "clique", len(c), "exact", er, flush=True)
