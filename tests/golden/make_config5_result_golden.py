#!/usr/bin/env python3
"""Generates tests/golden/config5_result_golden.json: the ORACLE's result on BASELINE config 5 (the 3DMatch pair
of tests/golden/config5_clouds.npz): FPFH (radii 2 and 5 voxels) + mutual nearest neighbours by the features oracle,
then the registration oracle with examples/teaser_python_fpfh_icp/helpers.py:45-60's parameters.  Recorded:
the correspondence list's digest, edge count, maximum clique (size, members, whether it is UNIQUE), R, t, and the
rotation / translation inlier lists.  The GPU test compares against this file (and still re-runs the oracle).
Run from the repo root (CPU only, about a minute):  python tests/golden/make_config5_result_golden.py"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import features as F  # noqa: E402
from oracle import oracle  # noqa: E402

C5 = np.load(os.path.join(ROOT, "tests", "golden", "config5_clouds.npz"))
A, B, vox = C5["cloud_bin_0"], C5["cloud_bin_4"], float(C5["voxel_size"])
fa, _ = F.fpfh_features(A, 2 * vox, 5 * vox)
fb, _ = F.fpfh_features(B, 2 * vox, 5 * vox)
corr = F.match(fa, fb, crosscheck=True)
p = dict(noise_bound=vox, cbar2=1.0, estimate_scaling=0, rotation_gnc_factor=1.4, rotation_max_iterations=10000,
         rotation_cost_threshold=1e-16)
o = oracle.solve(A[corr[:, 0]].astype(np.float64).T, B[corr[:, 1]].astype(np.float64).T, **p)
doc = dict(
    note=__doc__.split("Run from")[0].strip(),
    points=[int(len(A)), int(len(B))], voxel=vox, correspondences=int(len(corr)),
    correspondences_sha256=hashlib.sha256(np.ascontiguousarray(corr, dtype=np.int32).tobytes()).hexdigest(),
    num_edges=int(o["num_edges"]), clique_exact_run=int(o["clique_exact_run"]),
    clique_size=int(len(o["max_clique"])), clique_unique=bool(o["clique_unique"]),
    max_clique=[int(v) for v in o["max_clique"]],
    rotation=[float(v) for v in np.asarray(o["rotation"]).ravel()],
    translation=[float(v) for v in np.asarray(o["translation"]).ravel()],
    rotation_inliers=[int(v) for v in o["rotation_inliers"]],
    translation_inliers=[int(v) for v in o["translation_inliers"]])
json.dump(doc, open(os.path.join(ROOT, "tests", "golden", "config5_result_golden.json"), "w"), indent=1)
print({k: doc[k] for k in ("correspondences", "num_edges", "clique_size", "clique_unique", "clique_exact_run")})
