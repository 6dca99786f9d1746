#!/usr/bin/env python3
"""Generates tests/golden/config_golden.json: ORACLE results for the full-size BASELINE configs the
GPU test-suite cannot afford to re-run on the CPU each time:

  * config 3 -- N = 50 000, 99 % outliers (seeds 20250523 + 3 and + 2003): maximum clique, rotation /
    translation inlier index lists, R, t, GNC cost / iterations, edge count, and the SHA-256 of the
    whole 313 MB adjacency bitmap (so the GPU bitmap is compared bit for bit through its digest);
  * config 4 -- three elements (0, 57, 127) of the 128 x N = 5 000, 90 % outliers batch
    (seeds 20250523 + 4000 + b): the same fields;
  * config 2 -- N = 10 000, 95 % outliers (seed 20250523): the same fields (also solved live by the
    oracle in test_solve_parity_config2_10k; the fixture pins the oracle itself across rebuilds).

The oracle is the CPU restatement of the reference path (oracle/teaser_oracle.c), itself pinned to
the reference's golden vectors by tests/test_oracle_golden.py.  Minutes of CPU at N = 50 000, hence
a committed fixture (as tests/golden/scale_golden.json).  Run from the repo root:
    python tests/golden/make_config_golden.py
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

CASES = [
    ("config2", 20250523, 10000, 0.95),
    ("config3", 20250523 + 3, 50000, 0.99),
    ("config3_seed2", 20250523 + 3 + 2000, 50000, 0.99),   # (a second full-size case: round 6)
    ("config4_b0", 20250523 + 4000 + 0, 5000, 0.9),
    ("config4_b57", 20250523 + 4000 + 57, 5000, 0.9),
    ("config4_b127", 20250523 + 4000 + 127, 5000, 0.9),
]

# `make_config_golden.py name ...`: only those cases are recomputed, the others keep their committed entries
PATH = os.path.join(ROOT, "tests", "golden", "config_golden.json")
only = set(sys.argv[1:])
out = json.load(open(PATH)) if only and os.path.exists(PATH) else {}
for name, seed, n, rho in CASES:
    if only and name not in only:
        continue
    pr = tp.synth_problem(seed, n, rho, 0.01)
    t0 = time.time()
    o = oracle.solve(pr["src"], pr["dst"], **KW)
    t1 = time.time()
    _, bm = oracle.inlier_bitmap(pr["src"], pr["dst"], 0.01, 1.0, False)
    deg = np.unpackbits(bm.view(np.uint8), axis=1).sum(1).astype(np.int64)
    out[name] = {
        "seed": seed, "n": n, "outlier_ratio": rho, "noise_bound": 0.01,
        "valid": bool(o["valid"]), "clique_unique": bool(o["clique_unique"]),
        "clique_exact_run": bool(o["clique_exact_run"]), "max_core": int(o["max_core"]),
        "num_edges": int(o["num_edges"]),
        "max_clique": o["max_clique"].tolist(),
        "rotation_inliers": o["rotation_inliers"].tolist(),
        "translation_inliers": o["translation_inliers"].tolist(),
        "rotation": o["rotation"].reshape(-1).tolist(),
        "translation": o["translation"].tolist(),
        "gnc_cost": float(o["gnc_cost"]), "gnc_iterations": int(o["gnc_iterations"]),
        "bitmap_sha256": hashlib.sha256(np.ascontiguousarray(bm).tobytes()).hexdigest(),
        "degree_sum": int(deg.sum()), "degree_max": int(deg.max()),
        "degree_weighted_checksum": int((deg * (np.arange(n, dtype=np.int64) % 1009 + 1)).sum()),
        "oracle_solve_seconds": round(t1 - t0, 2),
    }
    print(name, {k: v for k, v in out[name].items()
                 if k not in ("max_clique", "rotation_inliers", "translation_inliers")}, flush=True)
json.dump(out, open(PATH, "w"), indent=0)
