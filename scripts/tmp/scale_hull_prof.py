import importlib, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
tp = importlib.import_module("teaser-plusplus_amd")
g = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests/golden/scale_golden.json')))[int(os.environ.get("GOLD", "0"))]
n, rho, k, nb = int(g["n"]), float(g["outlier_ratio"]), float(g["dst_scale"]), float(g["noise_bound"])
pr = tp.synth_problem(int(g["seed"]), n, rho, 0.01)
kw = dict(noise_bound=nb, cbar2=1.0, estimate_scaling=True, rotation_gnc_factor=1.4, rotation_max_iterations=100, rotation_cost_threshold=0.005)
tp.set_option("scale_hull", int(os.environ.get("HULL", "45")))
s = tp.RobustRegistrationSolver(tp.RobustRegistrationSolver.Params(**kw))
for rep in range(4):
    sol = s.solve(pr["src"], pr["dst"] * k)
print(sol.scale)
