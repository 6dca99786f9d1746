import importlib, sys, json, time
import numpy as np
sys.path.insert(0, '.')
tp = importlib.import_module("teaser-plusplus_amd")
kw = dict(noise_bound=0.01, cbar2=1.0, estimate_scaling=False, rotation_gnc_factor=1.4,
          rotation_max_iterations=100, rotation_cost_threshold=0.005)
import os
pr = tp.synth_problem(20250523 + int(os.environ.get("SEEDOFF", "0")), 50000, 0.99, 0.01)
tp.set_option("spec_bounds", 0)
for hb in (0, 1, 2, 4, 8):
    tp.set_option("heu_blocks", hb)
    tp.set_option("k4_debug", 1 if hb == 0 else 0)
    s = tp.RobustRegistrationSolver(tp.RobustRegistrationSolver.Params(**kw))
    s.solve(pr["src"], pr["dst"])
    tp.set_option("k4_debug", 0)
    s.set_profiling(1)
    hs = []
    for rep in range(3):
        s.solve(pr["src"], pr["dst"])
        pf = s.get_profile()
        hs.append((round(pf["heuristic_ms"], 4), round(pf["colour_ms"], 4), round(pf["total_ms"], 4)))
    print("heu_blocks", hb, hs, flush=True)
deg = s.getDegrees()
inl = pr["inliers"].astype(bool)
print("inlier deg min/med/max", int(deg[inl].min()), int(np.median(deg[inl])), int(deg[inl].max()), "outlier deg med/99/max",
      int(np.median(deg[~inl])), int(np.percentile(deg[~inl], 99)), int(deg[~inl].max()),
      "outliers above inlier median", int((deg[~inl] > np.median(deg[inl])).sum()), "above inlier min", int((deg[~inl] > deg[inl].min()).sum()))
