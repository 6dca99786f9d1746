#!/usr/bin/env python3
"""Wall time of the correspondence front-end calls at BASELINE config 5 (tests/golden/config5_clouds.npz),
without a profiler attached: FPFH of both clouds, the matcher.  GPU only."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tp = importlib.import_module("teaser-plusplus_amd")

C5 = np.load(os.path.join(ROOT, "tests", "golden", "config5_clouds.npz"))
A, B, vox = C5["cloud_bin_0"].astype(np.float32), C5["cloud_bin_4"].astype(np.float32), float(C5["voxel_size"])
est, m = tp.FPFHEstimation(), tp.Matcher()
fa = est.computeFPFHFeatures(A, 2 * vox, 5 * vox)
fb = est.computeFPFHFeatures(B, 2 * vox, 5 * vox)
m.calculateCorrespondences(A, B, fa, fb, False, True, False, 0)
ta, tb, tm = [], [], []
for _ in range(9):
    t0 = time.perf_counter()
    fa = est.computeFPFHFeatures(A, 2 * vox, 5 * vox)
    t1 = time.perf_counter()
    fb = est.computeFPFHFeatures(B, 2 * vox, 5 * vox)
    t2 = time.perf_counter()
    corr = m.calculateCorrespondences(A, B, fa, fb, False, True, False, 0)
    t3 = time.perf_counter()
    ta.append(t1 - t0), tb.append(t2 - t1), tm.append(t3 - t2)
med = lambda v: round(1e3 * float(np.median(v)), 3)
print(json.dumps(dict(probe="frontend", points=[len(A), len(B)], correspondences=len(corr), fpfh_a_ms=med(ta),
                      fpfh_b_ms=med(tb), match_ms=med(tm), total_ms=med(np.array(ta) + np.array(tb) + np.array(tm)))))
