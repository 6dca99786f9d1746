import importlib, sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
tp = importlib.import_module("teaser-plusplus_amd")
P = tp.RobustRegistrationSolver.Params(noise_bound=0.01, cbar2=1.0, estimate_scaling=False, rotation_gnc_factor=1.4,
                                       rotation_max_iterations=100, rotation_cost_threshold=0.005)
bad = 0
for n, rho, seed in [(8192, 0.99, 1), (8193, 0.99, 2), (8200, 0.988, 3), (8191, 0.99, 4), (65536, 0.99, 5), (65535, 0.99, 6), (65000, 0.992, 7), (12345, 0.99, 8), (16384, 0.99, 9), (16385, 0.991, 10)]:
    pr = tp.synth_problem(777 + seed, n, rho, 0.01)
    got = {}
    for mode in (0, 4096):
        tp.set_option("colour_mis", mode)
        tp.set_option("k4_debug", 1 if mode else 0)
        s = tp.RobustRegistrationSolver(P)
        sol = s.solve(pr["src"], pr["dst"]); raw = s.raw_solution()
        got[mode] = (s.getInlierMaxClique(), sol.rotation.copy(), sol.translation.copy(), int(raw.colour_uncoloured), int(raw.clique_exact_run))
    tp.set_option("k4_debug", 0)
    a, b = got[0], got[4096]
    ok = a[0] == b[0] and (a[1] == b[1]).all() and (a[2] == b[2]).all() and a[4] == b[4]
    bad += 0 if ok else 1
    print(n, rho, "X", a[3], b[3], "exact", a[4], b[4], "clique", len(a[0]), len(b[0]), int(pr["inliers"].sum()), "OK" if ok else "DIFF", flush=True)
print("differences:", bad)
