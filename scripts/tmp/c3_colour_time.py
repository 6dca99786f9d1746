import importlib, sys, os, json, time
sys.path.insert(0, '.')
tp = importlib.import_module("teaser-plusplus_amd")
kw = dict(noise_bound=0.01, cbar2=1.0, estimate_scaling=False, rotation_gnc_factor=1.4,
          rotation_max_iterations=100, rotation_cost_threshold=0.005)
pr = tp.synth_problem(777, 50000, 0.99, 0.01)
tp.set_option("colour_persistent", int(os.environ.get("PERSIST", "0")))
s = tp.RobustRegistrationSolver(tp.RobustRegistrationSolver.Params(**kw))
s.solve(pr["src"], pr["dst"])
s.set_profiling(1)
ts = []
for rep in range(5):
    s.solve(pr["src"], pr["dst"]); pf = s.get_profile()
    ts.append((round(pf["colour_ms"], 3), round(pf["total_ms"], 3)))
r = s.raw_solution()
print(os.environ.get("TAG", ""), "x", r.colour_uncoloured, "clique", r.clique_size, "(colour, total) ms", ts, flush=True)
