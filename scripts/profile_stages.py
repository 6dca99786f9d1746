#!/usr/bin/env python3
"""Per-stage device times (HIP events on the solver's stream) for a few workloads; GPU only."""
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tp = importlib.import_module("teaser-plusplus_amd")


def run(n, rho, batch, reps=5, seed=20250523):
    P = tp.RobustRegistrationSolver.Params(noise_bound=0.01, cbar2=1.0, estimate_scaling=False,
                                           rotation_gnc_factor=1.4, rotation_max_iterations=100,
                                           rotation_cost_threshold=0.005)
    s = tp.RobustRegistrationSolver(P)
    probs = [tp.synth_problem(seed + b, n, rho, 0.01) for b in range(batch)]
    srcs, dsts = [p["src"] for p in probs], [p["dst"] for p in probs]
    s.solve_batch(srcs, dsts)
    s.set_profiling(True)
    walls, profs = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        sols = s.solve_batch(srcs, dsts)
        walls.append(time.perf_counter() - t0)
        profs.append(s.get_profile())
    s.set_profiling(False)
    w2 = []
    for _ in range(reps):
        t0 = time.perf_counter()
        sols = s.solve_batch(srcs, dsts)
        w2.append(time.perf_counter() - t0)
    med = {k: float(np.median([p[k] for p in profs])) for k in profs[0]}
    raw = s.raw_solution(0)
    out = dict(n=n, rho=rho, batch=batch, wall_ms_profiled=1e3 * float(np.median(walls)),
               wall_ms=1e3 * float(np.median(w2)), clique=raw.clique_size, heu=raw.heuristic_size,
               exact=raw.clique_exact_run, gnc_iters=raw.gnc_iterations, edges=raw.num_edges,
               valid=raw.valid, **{k: round(v, 4) for k, v in med.items()})
    print(json.dumps(out), flush=True)
    return s, probs


if __name__ == "__main__":
    cfgs = [(1889, 0.6, 1), (5000, 0.9, 1), (10000, 0.95, 1), (10000, 0.95, 16), (5000, 0.9, 128)]
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        cfgs = [(50000, 0.99, 1)]
    if len(sys.argv) > 1 and sys.argv[1] == "batch":
        cfgs = [(10000, 0.95, 64), (5000, 0.9, 128)]
    for n, rho, b in cfgs:
        run(n, rho, b, reps=3 if n >= 50000 else 5)
