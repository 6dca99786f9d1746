import importlib, sys, os, json, time
import numpy as np
sys.path.insert(0, '.')
tp = importlib.import_module("teaser-plusplus_amd")
kw = dict(noise_bound=0.01, cbar2=1.0, estimate_scaling=False, rotation_gnc_factor=1.4,
          rotation_max_iterations=100, rotation_cost_threshold=0.005)
for seed in (20250523, 777, 778):
    pr = tp.synth_problem(seed, 50000, 0.99, 0.01)
    res = {}
    for mode in (0, 8192):
        tp.set_option("colour_persistent", mode)
        s = tp.RobustRegistrationSolver(tp.RobustRegistrationSolver.Params(**kw))
        s.solve(pr["src"], pr["dst"])
        s.set_profiling(1)
        ts = []
        for rep in range(3):
            t0 = time.perf_counter(); s.solve(pr["src"], pr["dst"]); w = time.perf_counter() - t0
            pf = s.get_profile()
            ts.append((round(pf["heuristic_ms"], 3), round(pf["colour_ms"], 3), round(pf["total_ms"], 3), round(1e3 * w, 3)))
        r = s.raw_solution()
        res[mode] = (s.getInlierMaxClique(), int(r.colour_uncoloured), int(r.clique_exact_run))
        print(seed, "persistent" if mode else "launches  ", "x", r.colour_uncoloured, "exact", r.clique_exact_run, "clique", r.clique_size, "(heu, colour, total, wall) ms", ts, flush=True)
    assert res[0] == res[8192], "routes differ"
print("routes agree")
