#!/usr/bin/env python3
"""The workflow of the reference's examples/teaser_python_fpfh_icp (helpers.py:9-60, without Open3D): two
clouds -> FPFH features -> mutual nearest-neighbour correspondences -> TEASER++ registration -> optional DRS
certificate, everything on the MI355X.  Usage:

    python examples/teaser_python_fpfh.py [src.ply dst.ply] [--voxel 0.05] [--certify]

Without file arguments it runs BASELINE config 5 from tests/golden/config5_clouds.npz (the 3DMatch pair
cloud_bin_0 / cloud_bin_4 after a 0.05 voxel grid).  ASCII / binary little-endian PLY with float x y z."""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tp = importlib.import_module("teaser-plusplus_amd")


def read_ply_xyz(path):
    with open(path, "rb") as f:
        header = []
        while True:
            line = f.readline().decode("ascii", "replace").strip()
            header.append(line)
            if line == "end_header":
                break
        n = int([h.split()[2] for h in header if h.startswith("element vertex")][0])
        props = [h.split()[2] for h in header if h.startswith("property")]
        if any(h.startswith("format ascii") for h in header):
            data = np.loadtxt(f, max_rows=n, dtype=np.float64)
            return data[:, [props.index("x"), props.index("y"), props.index("z")]].astype(np.float32)
        dt = np.dtype([(p, "<f4") for p in props])  # float properties only
        data = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
        return np.stack([data["x"], data["y"], data["z"]], axis=1).astype(np.float32)


def voxel_downsample(points, voxel):
    key = np.floor(points / voxel).astype(np.int64)
    _, inv = np.unique(key, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    out = np.zeros((inv.max() + 1, 3))
    np.add.at(out, inv, points)
    return (out / np.bincount(inv)[:, None]).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("clouds", nargs="*")
    ap.add_argument("--voxel", type=float, default=0.05)
    ap.add_argument("--certify", action="store_true")
    a = ap.parse_args()
    if len(a.clouds) == 2:
        A = voxel_downsample(read_ply_xyz(a.clouds[0]), a.voxel)
        B = voxel_downsample(read_ply_xyz(a.clouds[1]), a.voxel)
    else:
        c5 = np.load(os.path.join(ROOT, "tests", "golden", "config5_clouds.npz"))
        A, B, a.voxel = c5["cloud_bin_0"], c5["cloud_bin_4"], float(c5["voxel_size"])
    vox = a.voxel
    t0 = time.perf_counter()
    est = tp.FPFHEstimation()
    fa = est.computeFPFHFeatures(A, 2 * vox, 5 * vox)   # helpers.py:9-18: radii 2 and 5 voxels
    fb = est.computeFPFHFeatures(B, 2 * vox, 5 * vox)
    corr = tp.Matcher().calculateCorrespondences(A, B, fa, fb, False, True, False, 0)   # helpers.py:27-43
    t1 = time.perf_counter()
    params = tp.RobustRegistrationSolver.Params(noise_bound=vox, cbar2=1.0, estimate_scaling=False,
                                                rotation_gnc_factor=1.4, rotation_max_iterations=10000,
                                                rotation_cost_threshold=1e-16)               # helpers.py:45-60
    solver = tp.RobustRegistrationSolver(params)
    sol = solver.solve_correspondences(A, B, corr)
    t2 = time.perf_counter()
    print("%d / %d points, %d correspondences, max clique %d" % (len(A), len(B), len(corr),
                                                                 len(solver.getInlierMaxClique())))
    print("front-end %.1f ms, registration %.1f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
    print("R =\n%s\nt = %s" % (sol.rotation, sol.translation))
    if a.certify:
        c = np.array(corr)
        inl = np.zeros(len(c), dtype=bool)
        inl[solver.getInlierMaxClique()] = True
        cert = tp.DRSCertifier(noise_bound=vox, cbar2=1.0, max_iterations=100)
        # the certifier works on the translation-free measurements of the clique's correspondences
        src = A[c[:, 0]].astype(np.float64).T
        dst = B[c[:, 1]].astype(np.float64).T - sol.translation.reshape(3, 1)
        res = cert.certify(sol.rotation, src[:, inl], dst[:, inl], np.ones(int(inl.sum())))
        print(res)


if __name__ == "__main__":
    main()
