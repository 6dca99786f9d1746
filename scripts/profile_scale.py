#!/usr/bin/env python3
"""estimate_scaling = true (the K7 scale stage, registration.cc:410-425): wall time per solve for
  (a) batches of small problems (one sorting workgroup each; TEASER_SCALE_BATCH=0 runs them one after the other),
  (b) single large problems (device-wide radix sort).  GPU only.  One JSON line per configuration.
usage: profile_scale.py [small|large|all]"""
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tp = importlib.import_module("teaser-plusplus_amd")


def run(n, batch, reps=3, seed=77):
    P = tp.RobustRegistrationSolver.Params(noise_bound=0.02, cbar2=1.0, estimate_scaling=True,
                                           rotation_gnc_factor=1.4, rotation_max_iterations=100,
                                           rotation_cost_threshold=0.005)
    s = tp.RobustRegistrationSolver(P)
    probs = [tp.synth_problem(seed + b, n, 0.8, 0.01) for b in range(batch)]
    srcs, dsts = [p["src"] for p in probs], [p["dst"] * 1.5 for p in probs]
    sols = s.solve_batch(srcs, dsts)
    s.set_profiling(True)
    walls, tim = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        sols = s.solve_batch(srcs, dsts)
        walls.append(time.perf_counter() - t0)
        tim.append(s.get_profile()["tim_graph_ms"])
    print(json.dumps(dict(n=n, batch=batch, scale_batch=os.environ.get("TEASER_SCALE_BATCH", "1"),
                          wall_ms=1e3 * float(np.median(walls)), scale_plus_graph_ms=float(np.median(tim)),
                          scale0=float(sols[0].scale), valid=int(sum(bool(o.valid) for o in sols)))), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("small", "all"):
        run(100, 64)
        run(200, 256)
        run(500, 128)
        run(724, 64)
    if what in ("large", "all"):
        run(2000, 1)
        run(10000, 1)
