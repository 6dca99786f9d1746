import importlib, sys, time, json
import numpy as np
sys.path.insert(0, '.')
tp = importlib.import_module("teaser-plusplus_amd")
kw = dict(noise_bound=0.01, cbar2=1.0, estimate_scaling=False, rotation_gnc_factor=1.4,
          rotation_max_iterations=100, rotation_cost_threshold=0.005)
for (n, rho, B) in ((10000, 0.95, 64), (5000, 0.9, 128), (10000, 0.95, 1)):
    probs = [tp.synth_problem(20250523 + i, n, rho, 0.01) for i in range(B)]
    s = tp.RobustRegistrationSolver(tp.RobustRegistrationSolver.Params(**kw))
    s.set_profiling(1)
    for rep in range(4):
        sols = s.solve_batch([p["src"] for p in probs], [p["dst"] for p in probs])
        pf = s.get_profile()
        marks = [int(s.raw_solution(b).colour_uncoloured) for b in range(B)]
        deg = s.getDegrees(0)
        inl = probs[0]["inliers"].astype(bool)
        print(json.dumps(dict(n=n, B=B, rep=rep, closed=sum(m == -2 for m in marks), heuristic_ms=round(pf["heuristic_ms"], 4),
                              peel_ms=round(pf["peel_ms"], 4), tim=round(pf["tim_graph_ms"], 4), total=round(pf["total_ms"], 4),
                              min_inl_deg=int(deg[inl].min()), max_out_deg=int(deg[~inl].max()), K=int(inl.sum()))), flush=True)
