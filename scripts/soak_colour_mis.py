import importlib, sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
tp = importlib.import_module("teaser-plusplus_amd")
P = tp.RobustRegistrationSolver.Params(noise_bound=0.01, cbar2=1.0, estimate_scaling=False, rotation_gnc_factor=1.4,
                                       rotation_max_iterations=100, rotation_cost_threshold=0.005)
rng = np.random.default_rng(7)
bad = 0; ran = 0
for t in range(40):
    n = int(rng.integers(9000, 40000)); rho = float(rng.uniform(0.975, 0.993)); seed = int(rng.integers(1, 1 << 30))
    pr = tp.synth_problem(seed, n, rho, 0.01)
    got = {}
    for mode in (0, 4096):
        tp.set_option("colour_mis", mode)
        s = tp.RobustRegistrationSolver(P)
        sol = s.solve(pr["src"], pr["dst"]); raw = s.raw_solution()
        got[mode] = (s.getInlierMaxClique(), sol.rotation.copy(), sol.translation.copy(), int(raw.colour_uncoloured), int(raw.clique_exact_run))
    a, b = got[0], got[4096]
    ok = a[0] == b[0] and (a[1] == b[1]).all() and (a[2] == b[2]).all() and a[4] == b[4]
    ran += 1 if b[3] >= 0 else 0
    if not ok: bad += 1
    print(t, n, round(rho, 4), "X", a[3], b[3], "exact", a[4], b[4], "clique", len(a[0]), len(b[0]), "OK" if ok else "DIFF", flush=True)
print("cases with the colouring stage:", ran, "differences:", bad)
