#!/usr/bin/env python3
"""How the reference's matcher fixture (matcher-test.cc:46-85, 189 cross-checked pairs) depends on the way the normals'
covariance is accumulated (CPU only; oracle/features_oracle.c, feat_estimate_normals):
  fused    PCL's raw-coordinate accumulators as an FMA-contracting (-march=native) build compiles them  -> 189 / 189
  unfused  the same with every operation rounded (a generic x86-64 build)                               -> 174 / 189
  centred  PCL >= 1.10's accumulation relative to the first neighbour                                   -> 156 / 189
  double   PCA of the neighbourhood in float64 (numpy), i.e. the mathematically "right" normals         -> 156 / 189
and the median angle between the float forms' normals and the float64 ones.  One JSON line per combination."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import features as F  # noqa: E402


def normals_double(P, r):
    from scipy.spatial import cKDTree
    P64 = P.astype(np.float64)
    out = np.full_like(P64, np.nan)
    for i, idx in enumerate(cKDTree(P64).query_ball_point(P64, r)):
        if len(idx) < 3:
            continue
        Q = P64[idx] - P64[idx].mean(0)
        n = np.linalg.eigh(Q.T @ Q)[1][:, 0]
        out[i] = -n if n @ (-P64[i]) < 0 else n
    return out.astype(np.float32)


def main():
    G = np.load(os.path.join(ROOT, "tests", "golden", "features_golden.npz"))
    O, S = G["matcher_object"], G["matcher_scene"]
    ref = [tuple(r) for r in G["matcher_matches"].tolist()]
    forms = {"fused": lambda P: F.estimate_normals(P, 0.02), "unfused": lambda P: F.estimate_normals(P, 0.02, fused=False),
             "centred": lambda P: F.estimate_normals(P, 0.02, centred=True), "double": lambda P: normals_double(P, 0.02)}
    nrm = {k: (f(O), f(S)) for k, f in forms.items()}
    feat = {k: (F.compute_fpfh(O, a, 0.04), F.compute_fpfh(S, b, 0.04)) for k, (a, b) in nrm.items()}
    ang = lambda a, b: np.degrees(np.arccos(np.clip(np.abs((a.astype(np.float64) * b.astype(np.float64)).sum(1)), 0, 1)))
    for k in ("fused", "unfused", "centred"):
        d = ang(nrm[k][1], nrm["double"][1])
        print(json.dumps(dict(form=k, scene_normal_angle_to_float64_deg=dict(median=float(np.nanmedian(d)),
                                                                            p95=float(np.nanpercentile(d, 95))))))
    for ko in feat:
        for ks in feat:
            m = [tuple(r) for r in F.match(feat[ko][0], feat[ks][1], True).tolist()]
            print(json.dumps(dict(object=ko, scene=ks, pairs=len(m), in_fixture=len(set(m) & set(ref)),
                                  equal_in_order=m == ref)))


if __name__ == "__main__":
    main()
