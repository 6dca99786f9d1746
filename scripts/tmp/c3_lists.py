import importlib, sys, os
sys.path.insert(0, '.')
tp = importlib.import_module("teaser-plusplus_amd")
kw = dict(noise_bound=0.01, cbar2=1.0, estimate_scaling=False, rotation_gnc_factor=1.4,
          rotation_max_iterations=100, rotation_cost_threshold=0.005)
pr = tp.synth_problem(20250523 + int(os.environ.get("SEEDOFF", "0")), 50000, 0.99, 0.01)
tp.set_option("spec_bounds", 0)
tp.set_option("k4_debug", 1)
s = tp.RobustRegistrationSolver(tp.RobustRegistrationSolver.Params(**kw))
s.solve(pr["src"], pr["dst"])
