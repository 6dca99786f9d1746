#!/usr/bin/env python3
"""Wall time of solve() with estimate_scaling=true (TRIMs -> radix sort -> sweep) vs false."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tp = importlib.import_module("teaser-plusplus_amd")
for n, rho in ((2000, 0.9), (5000, 0.9), (10000, 0.95), (20000, 0.975)):
    pr = tp.synth_problem(20250523 + n, n, rho, 0.01)
    dst = pr["dst"] * 1.3
    for es in (False, True):
        P = tp.RobustRegistrationSolver.Params(noise_bound=0.013, cbar2=1.0, estimate_scaling=es,
                                               rotation_gnc_factor=1.4, rotation_max_iterations=100,
                                               rotation_cost_threshold=0.005)
        s = tp.RobustRegistrationSolver(P)
        d = dst if es else pr["dst"]
        sol = s.solve(pr["src"], d)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); sol = s.solve(pr["src"], d); ts.append(time.perf_counter() - t0)
        print("n=%d estimate_scaling=%s: %.2f ms  scale=%.6f clique=%d valid=%d" % (
            n, es, 1e3 * min(ts), sol.scale, len(s.getInlierMaxClique()), sol.valid), flush=True)
