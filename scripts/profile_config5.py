#!/usr/bin/env python3
"""BASELINE config 5 (3DMatch pair, real FPFH correspondences: tests/golden/config5_clouds.npz) through the GPU
front-end and solve(): per-stage device times; the regime where the exact clique search (K4) runs.  GPU only.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel table."""
import importlib
import json
import os
import sys
import time

import numpy as np

if os.environ.get("TEASER_PROFILE_WATCHDOG"):  # diagnostics: dump the Python stack and exit after N seconds
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ["TEASER_PROFILE_WATCHDOG"]), exit=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tp = importlib.import_module("teaser-plusplus_amd")

C5 = np.load(os.path.join(ROOT, "tests", "golden", "config5_clouds.npz"))
A, B, vox = C5["cloud_bin_0"], C5["cloud_bin_4"], float(C5["voxel_size"])
est = tp.FPFHEstimation()
est.computeFPFHFeatures(A, 2 * vox, 5 * vox)
t0 = time.perf_counter()
fa = est.computeFPFHFeatures(A, 2 * vox, 5 * vox)
fb = est.computeFPFHFeatures(B, 2 * vox, 5 * vox)
corr = tp.Matcher().calculateCorrespondences(A, B, fa, fb, False, True, False, 0)
t1 = time.perf_counter()
p = dict(noise_bound=vox, cbar2=1.0, estimate_scaling=False, rotation_gnc_factor=1.4,
         rotation_max_iterations=10000, rotation_cost_threshold=1e-16,
         max_clique_time_limit=float(os.environ.get("TEASER_CLIQUE_LIMIT", "3600")))
s = tp.RobustRegistrationSolver(tp.RobustRegistrationSolver.Params(**p))
s.solve_correspondences(A, B, corr)
s.set_profiling(True)
walls, profs = [], []
for _ in range(5):
    t2 = time.perf_counter()
    sol = s.solve_correspondences(A, B, corr)
    walls.append(time.perf_counter() - t2)
    profs.append(s.get_profile())
raw = s.raw_solution()
med = {k: round(float(np.median([q[k] for q in profs])), 4) for k in profs[0]}
print(json.dumps(dict(config=5, points=[len(A), len(B)], correspondences=len(corr), clique=raw.clique_size,
                      heuristic=raw.heuristic_size, exact_run=raw.clique_exact_run, edges=raw.num_edges,
                      front_end_ms=round(1e3 * (t1 - t0), 3), solve_wall_ms=round(1e3 * float(np.median(walls)), 3),
                      **med)))

if len(sys.argv) > 1 and sys.argv[1] == "batch":
    # the bench's batched config-5 step: 64 perturbed copies of the pair, each with its own FPFH correspondences
    import argparse
    import bench
    solver, wl = bench.config5_workload(tp, argparse.Namespace(seed=20250523, noise_bound=vox), 0, 64, 1)
    src, dst = wl["pool"][0]
    offs, szs = wl["offsets"][0], wl["sizes"][0]
    srcs = [np.ascontiguousarray(src[o:o + m].T) for o, m in zip(offs, szs)]
    dsts = [np.ascontiguousarray(dst[o:o + m].T) for o, m in zip(offs, szs)]
    solver.solve_batch(srcs, dsts)
    solver.set_profiling(True)
    walls, profs = [], []
    for _ in range(5):
        t2 = time.perf_counter()
        out = solver.solve_batch(srcs, dsts)
        walls.append(time.perf_counter() - t2)
        profs.append(solver.get_profile())
    med = {k: round(float(np.median([q[k] for q in profs])), 4) for k in profs[0]}
    cl = [int(solver.raw_solution(b).clique_size) for b in range(len(srcs))]
    ex = [int(solver.raw_solution(b).clique_exact_run) for b in range(len(srcs))]
    print(json.dumps(dict(config="5 batched", problems=len(srcs), sizes=[int(szs.min()), int(szs.max())],
                          clique_min_max=[min(cl), max(cl)], exact_runs=sum(ex),
                          step_wall_ms=round(1e3 * float(np.median(walls)), 3), **med)))
