import importlib, sys, os, json, time
import numpy as np
sys.path.insert(0, '.')
tp = importlib.import_module("teaser-plusplus_amd")
gold = json.load(open('tests/golden/scale_golden.json'))
for gi, g in enumerate(gold):
    n, rho, k, nb = int(g["n"]), float(g["outlier_ratio"]), float(g["dst_scale"]), float(g["noise_bound"])
    pr = tp.synth_problem(int(g["seed"]), n, rho, 0.01)
    src, dst = pr["src"], pr["dst"] * k
    kw = dict(noise_bound=nb, cbar2=1.0, estimate_scaling=True, rotation_gnc_factor=1.4, rotation_max_iterations=100,
              rotation_cost_threshold=0.005)
    res = {}
    for hull in (0, 60):
        tp.set_option("scale_hull", hull)
        tp.set_option("k4_debug", 1 if hull else 0)
        s = tp.RobustRegistrationSolver(tp.RobustRegistrationSolver.Params(**kw))
        sol = s.solve(src, dst)
        tp.set_option("k4_debug", 0)
        s.set_profiling(1)
        ts = []
        for rep in range(3):
            sol = s.solve(src, dst); pf = s.get_profile(); ts.append(round(pf["tim_graph_ms"], 3))
        raw = s.raw_solution()
        res[hull] = (sol.scale, int(raw.num_edges), raw.clique_size)
        print("golden", gi, "n", n, "hull", hull, "scale %.16g" % sol.scale, "oracle %.16g" % g["oracle_scale"], "diff %.3g" % (sol.scale - g["oracle_scale"]),
              "edges", raw.num_edges, "oracle", g["oracle_edges"], "scale+graph ms", ts, flush=True)
