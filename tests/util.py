"""Shared test helpers (metric definitions follow the reference's test tools)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "teaser_golden.npz")


def golden():
    return np.load(GOLDEN)


def angular_error(R_exp, R_est):
    """test/test-tools/test_utils.h:92-94 (reference)."""
    c = (np.trace(R_exp.T @ R_est) - 1) / 2
    return abs(np.arccos(min(max(c, -1.0), 1.0)))


def is_clique(dense_adj, members):
    m = np.asarray(members)
    sub = dense_adj[np.ix_(m, m)]
    return bool((sub | np.eye(len(m), dtype=bool)).all())
