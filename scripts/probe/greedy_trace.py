import importlib, sys, os
sys.path.insert(0, os.getcwd())
tp = importlib.import_module("teaser-plusplus_amd")
tp.LIB_PATH = os.path.join(os.getcwd(), "scripts/probe/libteaser_hip_trace.so")
P = tp.RobustRegistrationSolver.Params(noise_bound=0.01, cbar2=1.0, estimate_scaling=False, rotation_gnc_factor=1.4,
                                       rotation_max_iterations=100, rotation_cost_threshold=0.005)
s = tp.RobustRegistrationSolver(P)
for n, rho in ((50000, 0.99), (10000, 0.95)):
    pr = tp.synth_problem(777, n, rho, 0.01)
    s.solve(pr["src"], pr["dst"]); s.solve(pr["src"], pr["dst"])
    print("n", n, "clique", len(s.getInlierMaxClique()), flush=True)
