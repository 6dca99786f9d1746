#!/usr/bin/env python3
"""Collects the reference's certifier fixtures (test/teaser/data/certification_{small,large}_instances, read by
test/teaser/certification-test.cc:135-320) into tests/golden/certifier_golden.npz.  Run in the container that
has /root/reference; the GPU box only sees the committed .npz."""
import os
import numpy as np

REF = "/root/reference/test/teaser/data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "certifier_golden.npz")


def csv(path):
    return np.atleast_2d(np.loadtxt(path, delimiter=",", dtype=np.float64))


def params(path):
    d = {}
    for line in open(path):
        if ":" in line:
            k, v = line.split(":")
            d[k.strip()] = float(v)
    return d


out = {}
for kind, cases, names in (
        ("small", (1, 2, 3), ("A_inv", "M_affine_1st_iter", "Q_cost", "R_est", "W_1st_iter", "W_dual_1st_iter",
                              "block_diag_omega", "lambda_bar_init", "mu", "omega", "q_est",
                              "suboptimality_1st_iter", "suboptimality_traj", "theta_est", "v1", "v2")),
        ("large", (1, 2), ("R_est", "q_est", "suboptimality_1st_iter", "suboptimality_traj", "theta_est", "v1",
                           "v2"))):
    for c in cases:
        d = os.path.join(REF, "certification_%s_instances" % kind, "case_%d" % c)
        for n in names:
            out["%s%d_%s" % (kind, c, n)] = csv(os.path.join(d, n + ".csv"))
        p = params(os.path.join(d, "parameters.txt"))
        out["%s%d_params" % (kind, c)] = np.array([p.get("noise_bound", 0.01), p.get("cbar2", 1.0),
                                                   p.get("max_iterations", 200.0)])
np.savez_compressed(OUT, **out)
print(OUT, os.path.getsize(OUT), "bytes;", len(out), "arrays")
