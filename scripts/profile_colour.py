#!/usr/bin/env python3
"""Colouring bound of one large problem on both routes (colour_mis = 0 / on): stage times, |X|, k4_debug rounds; GPU only."""
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tp = importlib.import_module("teaser-plusplus_amd")
cfgs = [(50000, 0.99, 43), (50000, 0.99, 44), (30000, 0.99, 42), (20000, 0.985, 41), (10000, 0.97, 40)]
if len(sys.argv) > 1: cfgs = cfgs[:int(sys.argv[1])]
P = tp.RobustRegistrationSolver.Params(noise_bound=0.01, cbar2=1.0, estimate_scaling=False, rotation_gnc_factor=1.4,
                                       rotation_max_iterations=100, rotation_cost_threshold=0.005)
for n, rho, seed in cfgs:
    pr = tp.synth_problem(20250523 + seed, n, rho, 0.01)
    for mode in (0, 4096):
        tp.set_option("colour_mis", mode)
        s = tp.RobustRegistrationSolver(P)
        s.solve(pr["src"], pr["dst"])
        if mode:
            tp.set_option("k4_debug", 1); s.solve(pr["src"], pr["dst"]); tp.set_option("k4_debug", 0)
        s.set_profiling(True)
        profs = []
        for _ in range(5):
            s.solve(pr["src"], pr["dst"]); profs.append(s.get_profile())
        s.set_profiling(False)
        raw = s.raw_solution()
        med = {k: round(float(np.median([p[k] for p in profs])), 4) for k in profs[0]}
        print(json.dumps(dict(n=n, rho=rho, seed=seed, colour_mis=mode, X=int(raw.colour_uncoloured), clique=int(raw.clique_size),
                              exact=int(raw.clique_exact_run), valid=bool(raw.valid), **med)), flush=True)
