import importlib, json, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
tp = importlib.import_module("teaser-plusplus_amd")
P = tp.RobustRegistrationSolver.Params(noise_bound=0.01, cbar2=1.0, estimate_scaling=False, rotation_gnc_factor=1.4,
                                       rotation_max_iterations=100, rotation_cost_threshold=0.005)
for n in (2048, 3000, 4096, 6000, 8192, 12000):
    pr = tp.synth_problem(4242 + n, n, 0.99, 0.01)
    for mode in (0, 1024):
        tp.set_option("colour_mis", mode)
        s = tp.RobustRegistrationSolver(P)
        s.solve(pr["src"], pr["dst"])
        s.set_profiling(True)
        profs = []
        for _ in range(7):
            s.solve(pr["src"], pr["dst"]); profs.append(s.get_profile())
        s.set_profiling(False)
        raw = s.raw_solution()
        med = {k: round(float(np.median([p[k] for p in profs])), 4) for k in ("colour_ms", "total_ms", "exact_ms")}
        print(n, "mis" if mode else "old", "X", int(raw.colour_uncoloured), "clique", int(raw.clique_size), med, flush=True)
tp.set_option("colour_mis", 8192)
