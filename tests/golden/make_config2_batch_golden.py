#!/usr/bin/env python3
"""Generates tests/golden/config2_batch_golden.json: ORACLE results for a WHOLE headline batch -- 64 problems of
N = 10 000, 95 % outliers (seeds 20250523 + b, b = 0 .. 63: the shape bench.py submits per step) -- so that the route
the benchmark times (teaser_hip_submit_batch / teaser_hip_wait, two batches in flight, finisher threads) is compared with
the oracle problem by problem, not only with the synchronous route.

Per problem: the maximum clique, the rotation / translation inlier lists (as SHA-256 of their int32 bytes plus their
lengths: 64 x 3 lists of ~500 indices would be 100 k numbers), R, t, the edge count, whether the oracle found the
maximum clique unique, and the SHA-256 of the 12.5 MB adjacency bitmap.  A few minutes of CPU.  Run from the repo root:
    python tests/golden/make_config2_batch_golden.py
"""
import hashlib
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
tp = importlib.import_module("teaser-plusplus_amd")
from oracle import oracle  # noqa: E402

KW = dict(noise_bound=0.01, cbar2=1.0, estimate_scaling=0, rotation_gnc_factor=1.4,
          rotation_max_iterations=100, rotation_cost_threshold=0.005)
B, N, RHO, SEED = 64, 10000, 0.95, 20250523


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(np.asarray(a, dtype=np.int32)).tobytes()).hexdigest()


out = {"batch": B, "n": N, "outlier_ratio": RHO, "noise_bound": 0.01, "seed0": SEED, "problems": []}
t0 = time.time()
for b in range(B):
    pr = tp.synth_problem(SEED + b, N, RHO, 0.01)
    o = oracle.solve(pr["src"], pr["dst"], **KW)
    _, bm = oracle.inlier_bitmap(pr["src"], pr["dst"], 0.01, 1.0, False)
    out["problems"].append({
        "seed": SEED + b, "valid": bool(o["valid"]), "clique_unique": bool(o["clique_unique"]),
        "num_edges": int(o["num_edges"]), "clique_size": int(len(o["max_clique"])),
        "max_clique_sha256": sha(o["max_clique"]),
        "n_rotation_inliers": int(len(o["rotation_inliers"])), "rotation_inliers_sha256": sha(o["rotation_inliers"]),
        "n_translation_inliers": int(len(o["translation_inliers"])),
        "translation_inliers_sha256": sha(o["translation_inliers"]),
        "rotation": o["rotation"].reshape(-1).tolist(), "translation": o["translation"].tolist(),
        "bitmap_sha256": hashlib.sha256(np.ascontiguousarray(bm).tobytes()).hexdigest(),
    })
    print("problem %d: clique %d unique %s edges %d  (%.0f s)" % (b, len(o["max_clique"]), o["clique_unique"],
                                                                   o["num_edges"], time.time() - t0), flush=True)
path = os.path.join(ROOT, "tests", "golden", "config2_batch_golden.json")
json.dump(out, open(path, "w"), indent=0)
print("wrote", path)
