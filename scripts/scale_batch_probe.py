#!/usr/bin/env python3
"""estimate_scaling = true on batches of mid-size problems: one problem alone, 64 one after the other
(TEASER_SCALE_MID_BATCH=0) and 64 through the shared sort.  GPU only.   scale_batch_probe.py [n] [batch]"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tp = importlib.import_module("teaser-plusplus_amd")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nb = 0.01
probs = [tp.synth_problem(900 + i, n, 0.6, nb) for i in range(B)]
src = [q["src"] for q in probs]
dst = [q["dst"] * (0.8 + 0.01 * i) for i, q in enumerate(probs)]
P = tp.RobustRegistrationSolver.Params(noise_bound=nb * 1.5, cbar2=1.0, estimate_scaling=True, rotation_gnc_factor=1.4,
                                       rotation_max_iterations=100, rotation_cost_threshold=1e-12)
s = tp.RobustRegistrationSolver(P)


def timed(fn, reps=5):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(t))


one_ms = timed(lambda: s.solve(src[0], dst[0]))
batch_ms = timed(lambda: s.solve_batch(src, dst), 3)
print(json.dumps(dict(probe="scale_batch", n=n, batch=B, mid_batch=os.environ.get("TEASER_SCALE_MID_BATCH", "1"),
                      one_problem_ms=round(one_ms, 3), batch_ms=round(batch_ms, 3),
                      per_problem_in_batch_ms=round(batch_ms / B, 4), ratio_batch_over_one=round(batch_ms / one_ms, 2))))
