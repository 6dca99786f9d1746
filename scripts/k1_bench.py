"""K1 (tim_graph_kernel) timing via the solver's per-stage HIP events."""
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tp = importlib.import_module("teaser-plusplus_amd")

def run(n, rho, batch, reps=5):
    P = tp.RobustRegistrationSolver.Params(noise_bound=0.01, cbar2=1.0, estimate_scaling=False,
                                           rotation_gnc_factor=1.4, rotation_max_iterations=100,
                                           rotation_cost_threshold=0.005,
                                           inlier_selection_mode=tp.InlierSelectionMode.PMC_HEU)
    s = tp.RobustRegistrationSolver(P)
    probs = [tp.synth_problem(1000 + b, n, rho, 0.01) for b in range(batch)]
    srcs, dsts = [p["src"] for p in probs], [p["dst"] for p in probs]
    s.solve_batch(srcs, dsts)
    s.set_profiling(True)
    t = []
    for _ in range(reps):
        s.solve_batch(srcs, dsts)
        t.append(s.get_profile()["tim_graph_ms"])
    pairs = batch * n * (n - 1) / 2
    ms = float(np.median(t))
    print(json.dumps(dict(n=n, batch=batch, k1_ms=round(ms, 4), gpairs_s=round(pairs / ms / 1e6, 1),
                          tflops20=round(20 * pairs / ms / 1e9, 2))), flush=True)

if __name__ == "__main__":
    for n, b in [(10000, 1), (10000, 16), (20000, 4), (50000, 1)]:
        run(n, 0.95 if n < 50000 else 0.99, b)
