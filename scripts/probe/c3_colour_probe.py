"""config 3 (N = 50 000, 99 % outliers): how many survivors the colouring bound leaves uncoloured (|X|) and what the
stage costs, per seed."""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
tp = importlib.import_module("teaser-plusplus_amd")
P = tp.RobustRegistrationSolver.Params(noise_bound=0.01, cbar2=1.0, estimate_scaling=False, rotation_gnc_factor=1.4,
                                       rotation_max_iterations=100, rotation_cost_threshold=0.005)
s = tp.RobustRegistrationSolver(P)
for seed in (777, 778, 779):
    pr = tp.synth_problem(seed, 50000, 0.99, 0.01)
    s.solve(pr["src"], pr["dst"])
    s.set_profiling(1)
    s.solve(pr["src"], pr["dst"])
    pf = s.get_profile()
    s.set_profiling(0)
    r = s.raw_solution()
    print(json.dumps(dict(seed=seed, clique=r.clique_size, x_count=r.colour_uncoloured, exact_run=r.clique_exact_run,
                          colour_ms=round(pf["colour_ms"], 4), peel_ms=round(pf["peel_ms"], 4), exact_ms=round(pf["exact_ms"], 4),
                          heuristic_ms=round(pf["heuristic_ms"], 4), total_ms=round(pf["total_ms"], 4))), flush=True)
