#!/usr/bin/env python3
"""Generates tests/golden/scale_golden.json: the ORACLE's TLS scale estimate and edge count for a
full-size estimate_scaling=true problem (N = 10 000, 95 % outliers, dst scaled by 1.3) -- 43 s of
single-threaded CPU, too slow for the test suite, so the result is committed as a fixture.
The oracle itself is pinned against the reference's golden vectors (tests/test_oracle_golden.py).
Note: at 95 % outliers the reference's TLS scale estimate does NOT recover 1.3 (only 0.25 % of the
5e7 TRIMs are inlier pairs); the fixture pins what the algorithm returns, not the ground truth."""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
tp = importlib.import_module("teaser-plusplus_amd")
from oracle import oracle  # noqa: E402

out = []
for n, rho, scale in ((10000, 0.95, 1.3), (6000, 0.6, 0.8)):
    pr = tp.synth_problem(20250523 + n, n, rho, 0.01)
    dst = pr["dst"] * scale
    nb = 0.01 * scale
    sc, bm = oracle.inlier_bitmap(pr["src"], dst, nb, 1.0, True)
    out.append({"seed": 20250523 + n, "n": n, "outlier_ratio": rho, "dst_scale": scale,
                "noise_bound": nb, "oracle_scale": sc,
                "oracle_edges": int(np.unpackbits(bm.view(np.uint8)).sum()) // 2})
    print(out[-1], flush=True)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "scale_golden.json"), "w"), indent=1)
