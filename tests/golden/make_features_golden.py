#!/usr/bin/env python3
"""Generates tests/golden/features_golden.npz from the reference's own fixtures for the correspondence
front-end (FPFH + matcher):

  * test/teaser/data/bunny.pcd (397 points) and bunny_fpfh.csv (397 x 33): feature-test.cc:55-90 expects
    teaser::FPFHEstimation::computeFPFHFeatures(cloud, 0.03, 0.05) to reproduce the CSV to 1e-4;
  * test/teaser/data/matcher-test-object-1.ply (1000 points), matcher-test-scene-1.ply (60 865 points) and
    matcher-test-matches-1.csv (189 pairs, 1-based): matcher-test.cc:46-85 expects FPFH (0.02, 0.04) +
    Matcher::calculateCorrespondences(..., false, true, false, 0.95) to reproduce the pairs;
  * test/teaser/data/canstick.ply: matcher-test.cc:21-44 (self matching).

Run from the repo root (needs /root/reference):  python tests/golden/make_features_golden.py
"""
import os

import numpy as np

REF = "/root/reference/test/teaser/data/"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def read_ascii_ply(path):
    with open(path) as f:
        lines = f.read().split("\n")
    i = lines.index("end_header")
    n = int([l for l in lines if l.startswith("element vertex")][0].split()[-1])
    return np.array([[float(v) for v in l.split()[:3]] for l in lines[i + 1:i + 1 + n]], dtype=np.float32)


out = dict(
    bunny_pts=np.loadtxt(REF + "bunny.pcd", skiprows=10, dtype=np.float32),
    bunny_fpfh=np.loadtxt(REF + "bunny_fpfh.csv", dtype=np.float32).reshape(-1, 33),
    matcher_object=read_ascii_ply(REF + "matcher-test-object-1.ply"),
    matcher_scene=read_ascii_ply(REF + "matcher-test-scene-1.ply"),
    matcher_matches=(np.loadtxt(REF + "matcher-test-matches-1.csv", delimiter=",", dtype=np.int64) - 1).astype(np.int32),
    canstick=read_ascii_ply(REF + "canstick.ply"),
)
for k, v in out.items():
    print(k, v.shape, v.dtype)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "features_golden.npz"), **out)
