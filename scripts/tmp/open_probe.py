"""Which problems does the degree closure leave open on the headline workload, and why?  For each open problem: the
clique found, the inlier count, how many inliers are missing from the clique / outliers inside it, and the subgraph
induced by the inliers (missing edges)."""
import importlib, sys, json
import numpy as np
sys.path.insert(0, '.')
tp = importlib.import_module("teaser-plusplus_amd")
kw = dict(noise_bound=0.01, cbar2=1.0, estimate_scaling=False, rotation_gnc_factor=1.4,
          rotation_max_iterations=100, rotation_cost_threshold=0.005)
n, rho, B = 10000, 0.95, 64
tot_open = 0
for base in (20250523, 777000, 31337):
    probs = [tp.synth_problem(base + i, n, rho, 0.01) for i in range(B)]
    s = tp.RobustRegistrationSolver(tp.RobustRegistrationSolver.Params(**kw))
    s.solve_batch([p["src"] for p in probs], [p["dst"] for p in probs])
    for b in range(B):
        r = s.raw_solution(b)
        if int(r.colour_uncoloured) == -2:
            continue
        tot_open += 1
        inl = probs[b]["inliers"]
        cl = np.array(s.getInlierMaxClique(b))
        K = int(inl.sum())
        in_cl = np.zeros(n, bool); in_cl[cl] = True
        # adjacency among inliers: recompute the TIM predicate in numpy
        src, dst = probs[b]["src"][:, inl], probs[b]["dst"][:, inl]
        A = np.linalg.norm(src[:, :, None] - src[:, None, :], axis=0)
        Bm = np.linalg.norm(dst[:, :, None] - dst[:, None, :], axis=0)
        adj = np.abs(A - Bm) <= 2 * 0.01
        np.fill_diagonal(adj, True)
        miss = int((~adj).sum() // 2)
        deg = s.getDegrees(b)
        print(json.dumps(dict(base=base, b=b, K=K, clique=len(cl), inliers_not_in_clique=int((inl & ~in_cl).sum()),
                              outliers_in_clique=int((~inl & in_cl).sum()), missing_inlier_edges=miss,
                              max_out_deg=int(deg[~inl].max()), min_inl_deg=int(deg[inl].min()))), flush=True)
print("open", tot_open, "of", 3 * B)
