"""The correspondence front-end oracle (oracle/features_oracle.c: PCL normals + FPFH, FLANN-style matcher)
against the reference's own fixtures for that path (tests/golden/features_golden.npz, made by
tests/golden/make_features_golden.py from test/teaser/data).

What is pinned and what is not.  The FPFH fixture (feature-test.cc:55-90, tolerance 1e-4) decides 71 000
histogram bins by floor() of float features; PCL evaluates those with libm's acosf / atan2f, whose last-bit
behaviour is not portable, while the restatement uses deterministic functions built from IEEE basic
operations (so that the GPU can reproduce it bit for bit).  Result: the f1 and f2 histograms (bins 0-21)
agree with the fixture for EVERY point; the f3 histogram agrees for >= 70 % of the points, the rest being
the neighbourhoods of three point pairs whose |angle1| == |angle2| tie (pfh_tools.hpp, "switch p1 and p2")
falls the other way -- a mirror image inside the f3 histogram, never a change of its total.  The matcher
fixture (matcher-test.cc:46-85) goes through those features: >= 90 % of the reference pairs are reproduced."""
import numpy as np
import pytest

from oracle import features as F
from util import ROOT

import os

G = np.load(os.path.join(ROOT, "tests", "golden", "features_golden.npz"))


def test_fpfh_bunny_fixture():
    f, nv = F.fpfh_features(G["bunny_pts"], 0.03, 0.05)
    exp = G["bunny_fpfh"]
    assert f.shape == exp.shape == (397, 33) and np.isfinite(f).all() and np.isfinite(nv).all()
    assert np.allclose(np.linalg.norm(nv, axis=1), 1.0, atol=1e-5)
    d = np.abs(f - exp)
    assert d[:, :22].max() < 2e-4                      # f1, f2: every point
    rows_ok = (d.max(1) < 2e-4).sum()
    assert rows_ok >= 0.70 * len(f), rows_ok           # f3: all but the tie neighbourhoods
    # a tie flips a pair's f3 sign: mass moves between mirror bins (22 + k <-> 32 - k), totals unchanged
    bad = np.flatnonzero(d.max(1) >= 2e-4)
    diff = (f - exp)[bad, 22:]
    assert np.abs(diff + diff[:, ::-1]).max() < 5e-4
    for g in range(3):  # every 11-bin histogram sums to 100
        assert np.abs(f[:, 11 * g:11 * g + 11].sum(1) - 100).max() < 1e-3


def test_normals_are_pca_normals_oriented_to_the_origin():
    pts = G["bunny_pts"].astype(np.float64)
    nv = F.estimate_normals(G["bunny_pts"], 0.03).astype(np.float64)
    from scipy.spatial import cKDTree
    t = cKDTree(pts)
    worst = 0.0
    for i in range(0, len(pts), 7):
        idx = t.query_ball_point(pts[i], 0.03)
        w, v = np.linalg.eigh(np.cov(pts[idx].T, bias=True))
        worst = max(worst, np.degrees(np.arccos(min(1.0, abs(v[:, 0] @ nv[i])))))
        assert (-pts[i]) @ nv[i] >= -1e-9  # flipNormalTowardsViewpoint, viewpoint (0, 0, 0)
    assert worst < 0.1  # float covariance of raw coordinates, as PCL accumulates it: ~0.03 degrees


def test_matcher_self_matching():
    """matcher-test.cc:21-44: a cloud against itself -> the identity correspondences."""
    pts = G["canstick"]
    f, _ = F.fpfh_features(pts, 0.03, 0.05)
    m = F.match(f, f, crosscheck=True)
    # points with identical descriptors (symmetric surface) resolve to the lowest index on both sides, so
    # the cross check keeps exactly the points that are their own nearest neighbour
    assert (m[:, 0] == m[:, 1]).all() and len(m) >= 0.9 * len(pts)


@pytest.mark.slow
def test_matcher_case1_fixture():
    """matcher-test.cc:46-85 (object 1000 points, scene 60 865 points; minutes of brute force on the CPU)."""
    fo, _ = F.fpfh_features(G["matcher_object"], 0.02, 0.04)
    fs, _ = F.fpfh_features(G["matcher_scene"], 0.02, 0.04)
    m = F.match(fo, fs, crosscheck=True)
    ref = set(map(tuple, G["matcher_matches"].tolist()))
    got = set(map(tuple, m.tolist()))
    assert len(ref & got) >= 0.9 * len(ref) and abs(len(got) - len(ref)) <= 0.1 * len(ref)


def test_match_small_bruteforce_properties():
    rng = np.random.default_rng(3)
    a = rng.normal(size=(300, 33)).astype(np.float32)
    b = np.concatenate([a[:200] + 0.01 * rng.normal(size=(200, 33)).astype(np.float32),
                        rng.normal(size=(150, 33)).astype(np.float32)])
    m = F.match(a, b, crosscheck=True)
    assert set(map(tuple, m.tolist())) >= {(i, i) for i in range(200)}
    m2 = F.match(a, b, crosscheck=False)  # corres_ij + corres_ji, sorted, unique (matcher.cc:187-191, 295-296)
    assert len(m2) >= len(m) and (np.diff(m2[:, 0]) >= 0).all()
    sw = F.match(b, a, crosscheck=True)   # the larger cloud is searched first; pairs stay (src, dst)
    assert set(map(tuple, sw[:, ::-1].tolist())) == set(map(tuple, m.tolist()))
