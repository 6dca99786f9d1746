#!/usr/bin/env python3
"""k4_debug lines of the exact search on bench.py's `configs.scale` problem 0 (N = 10 k, estimate_scaling = true)."""
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
tp = importlib.import_module("teaser-plusplus_amd")
g = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "scale_golden.json")))[0]
pr = tp.synth_problem(int(g["seed"]), int(g["n"]), float(g["outlier_ratio"]), 0.01)
P = tp.RobustRegistrationSolver.Params(noise_bound=float(g["noise_bound"]), cbar2=1.0, estimate_scaling=True, rotation_gnc_factor=1.4,
                                       rotation_max_iterations=100, rotation_cost_threshold=0.005)
s = tp.RobustRegistrationSolver(P)
s.solve(pr["src"], pr["dst"] * float(g["dst_scale"]))
tp.set_option("k4_debug", 1)
s.set_profiling(True)
sol = s.solve(pr["src"], pr["dst"] * float(g["dst_scale"]))
print(json.dumps({k: round(float(v), 4) for k, v in s.get_profile().items() if k.endswith("_ms")}))
print(len(s.getInlierMaxClique()), sol.scale)
