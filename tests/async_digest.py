#!/usr/bin/env python3
"""Helper of test_async_paths_agree_across_host_modes (GPU): batches that alternate between "every bound closed by the
peel" and "some problems need the colouring bound / exact search" through the asynchronous API at depth 3, and one
digest over everything the solver returns.  The environment (TEASER_HIP_FINISHER, TEASER_HIP_SPEC_BOUNDS) selects the
host path; the digest must not depend on it.  Prints one JSON line."""
import hashlib
import importlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tp = importlib.import_module("teaser-plusplus_amd")
from util import HipBuffers  # noqa: E402


def packed(probs):
    src = np.ascontiguousarray(np.concatenate([p["src"].T for p in probs], axis=0))
    dst = np.ascontiguousarray(np.concatenate([p["dst"].T for p in probs], axis=0))
    n = np.array([p["src"].shape[1] for p in probs], dtype=np.int32)
    off = np.concatenate([[0], np.cumsum(n)[:-1]]).astype(np.int64)
    return src, dst, off, n


def main():
    # (n, outlier ratio): 0.8 -> the peel closes the greedy bound; 0.99 at n >= 12000 -> outliers have more neighbours
    # than the clique has members, the peel leaves the problem open and the colouring bound has to run
    plan = [
        [(1500, 0.8), (700, 0.8), (64, 0.8)],                     # all closed
        [(12000, 0.99), (900, 0.8), (16000, 0.99)],               # open + closed: speculation switches ON behind it
        [(2000, 0.8), (14000, 0.99)],                             # speculative: one proven, one open
        [(800, 0.8), (333, 0.8), (1200, 0.8), (50, 0.8)],         # speculative on an all-closed batch: switches OFF
        [(1000, 0.8), (1, 0.0), (600, 0.8)],                      # not speculative again
        [(13000, 0.99), (15000, 0.99)],                           # open, not speculative
        [(12500, 0.99), (300, 0.8)],                              # speculative
    ]
    batches = [[tp.synth_problem(4000 + 37 * k + i, n, rho, 0.01) for i, (n, rho) in enumerate(b)] for k, b in enumerate(plan)]
    P = tp.RobustRegistrationSolver.Params(noise_bound=0.01, cbar2=1.0, estimate_scaling=False, rotation_gnc_factor=1.4,
                                           rotation_max_iterations=100, rotation_cost_threshold=0.005)
    s = tp.RobustRegistrationSolver(P)
    s.set_pipeline_depth(3)
    mem = HipBuffers()
    h = hashlib.sha256()
    coloured = 0
    exact = 0
    for host in (False, True):
        put = mem.pinned if host else mem.device
        args = []
        for probs in batches:
            src, dst, off, n = packed(probs)
            args.append((put(src), put(dst), off, n))
        tickets = []
        order = []
        for k, a in enumerate(args):
            if len(tickets) == 3:
                t, kk = tickets.pop(0)
                order.append((kk, s.wait(t)))
                digest_batch(s, order[-1][1], len(batches[kk]), h)
            tickets.append((s.submit_batch(a[0], a[1], a[2], a[3], host=host), k))
        while tickets:
            t, kk = tickets.pop(0)
            out = s.wait(t)
            digest_batch(s, out, len(batches[kk]), h)
            order.append((kk, out))
        for kk, out in order:
            for b in range(len(batches[kk])):
                coloured += int(out[b].colour_uncoloured >= 0)
                exact += int(out[b].clique_exact_run)
    print(json.dumps(dict(digest=h.hexdigest(), coloured=coloured, exact=exact,
                          finisher=os.environ.get("TEASER_HIP_FINISHER", "1"),
                          spec=os.environ.get("TEASER_HIP_SPEC_BOUNDS", "1"))))


def digest_batch(s, out, B, h):
    for b in range(B):
        o = out[b]
        h.update(np.array([o.valid, o.status, o.n, o.clique_size, o.n_rotation_inliers, o.n_translation_inliers,
                           o.heuristic_size], dtype=np.int64).tobytes())
        h.update(np.array(o.rotation[:], dtype=np.float64).tobytes())
        h.update(np.array(o.translation[:], dtype=np.float64).tobytes())
        h.update(np.array(s.getInlierMaxClique(b), dtype=np.int64).tobytes())
        if o.valid:
            h.update(np.array(s.getRotationInliers(b), dtype=np.int64).tobytes())
            h.update(np.array(s.getTranslationInliers(b), dtype=np.int64).tobytes())


if __name__ == "__main__":
    main()
