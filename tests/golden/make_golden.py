#!/usr/bin/env python3
"""Collect the reference's own hot-path golden vectors into tests/golden/teaser_golden.npz.

Run in the build container (where /root/reference exists); the .npz travels with the repo, so
no test ever reads /root/reference at run time.  Nothing here is reference *source*: only the
data fixtures and the literal known answers the reference's tests assert, each with the
file:line it was taken from (paths relative to /root/reference).

    python tests/golden/make_golden.py [--reference /root/reference]
"""
import argparse
import os

import numpy as np


def read_csv_matrix(path):
    rows = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line:
                rows.append([float(t) for t in line.split(",") if t != ""])
    return np.array(rows, dtype=np.float64)


def read_ply_xyz_f32(path):
    """ASCII PLY vertex reader; returns float32 Nx3 exactly like teaser::PLYReader fills
    PointXYZ{float x,y,z} (teaser/src/ply_io.cc:28-80, geometry.h:15-23)."""
    with open(path, "rb") as f:
        header = []
        while True:
            line = f.readline().decode("ascii").strip()
            header.append(line)
            if line == "end_header":
                break
        assert any(h.startswith("format ascii") for h in header), path
        nv = [int(h.split()[-1]) for h in header if h.startswith("element vertex")][0]
        pts = np.empty((nv, 3), dtype=np.float32)
        for i in range(nv):
            tok = f.readline().split()
            # std::stof-like: parse as double then round to float32
            pts[i] = [np.float32(float(tok[0])), np.float32(float(tok[1])), np.float32(float(tok[2]))]
    return pts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "teaser_golden.npz"))
    args = ap.parse_args()
    ref = args.reference
    T = os.path.join(ref, "test/teaser/data/registration_test")
    B = os.path.join(ref, "test/benchmark/data")
    g = {}

    # --- scalar TLS known answers: test/teaser/tls-test.cc:25-84
    g["tls1_x"] = np.array([0.5, 1, 0.6, 0.7, 1.2])
    g["tls1_r"] = np.array([0.9, 0.9, 0.4, 0.5, 0.4])
    g["tls1_est"] = np.array(0.8383)
    g["tls1_mask"] = np.array([1, 1, 1, 1, 1], dtype=np.uint8)
    g["tls2_x"] = np.array([0.5, 1, 0.6, 0.7, 1.2, 10])
    g["tls2_r"] = np.array([0.9, 0.9, 0.4, 0.5, 0.4, 0.5])
    g["tls2_est"] = np.array(0.8383)
    g["tls2_mask"] = np.array([1, 1, 1, 1, 1, 0], dtype=np.uint8)
    g["tls3_x"] = np.array([0.5, 1, 0.6, 20, 16, 10])
    g["tls3_r"] = np.array([0.9, 0.9, 0.4, 0.5, 0.4, 0.5])
    g["tls3_est"] = np.array(0.6425)
    g["tls3_mask"] = np.array([1, 1, 1, 0, 0, 0], dtype=np.uint8)
    g["tls_tol"] = np.array(1e-3)  # tls-test.cc:45

    # --- translation known answer: test/teaser/translation-solver-test.cc:88-112
    g["trans_v1"] = read_csv_matrix(os.path.join(T, "translation_test_v1_inliers.csv"))  # 3x34
    g["trans_v2"] = read_csv_matrix(os.path.join(T, "translation_test_v2_inliers.csv"))
    g["trans_noise_bound"] = np.array(0.00673642835)
    g["trans_expected_t"] = np.array([-0.098430131086161, 0.008679113091532, 0.197317864174211])
    g["trans_tol"] = np.array(1e-5)

    # --- GNC-TLS rotation known answer: test/teaser/rotation-solver-test.cc:221-250
    g["rot_src"] = read_csv_matrix(os.path.join(T, "rotation_only_src.csv"))  # 200x3
    g["rot_expected_R"] = np.array(
        [[0.997379773225804, -0.019905935977315, -0.069551000516966],
         [0.013777311189888, 0.996068297974922, -0.087510750572249],
         [0.071019530105605, 0.086323226782879, 0.993732623426126]])
    g["rot_params"] = np.array([100, 1e-12, 1.4, 1e-3])  # max_iter, cost_thr, gnc_factor, noise_bound (:148)
    g["rot_tol"] = np.array(1e-5)
    # --- FGR rotation known answer: test/teaser/rotation-solver-test.cc:101-134 (same src and
    # expected_R; Params{max_iterations 1, cost_threshold 0.025, gnc_factor 1.4, noise_bound 1e-3}, :123)
    g["fgr_params"] = np.array([1, 0.025, 1.4, 1e-3])
    # --- QUATRO known answer: test/teaser/registration-test.cc:179-216 (yaw-only expected_R :204-208;
    # Params noise_bound 0.0067364, max_iterations 100, gnc 1.4, cost_threshold 0.005)
    g["quatro_expected_R"] = np.array([[0.997379773225804, -0.072343541246221, 0.0],
                                       [0.072343541246221, 0.997379773225804, 0.0],
                                       [0.0, 0.0, 1.0]])
    g["quatro_params"] = np.array([100, 0.005, 1.4, 0.0067364])

    # --- objectIn / sceneIn: test/teaser/registration-test.cc:256-392, scale-solver-test.cc
    g["object_in"] = read_csv_matrix(os.path.join(T, "objectIn.csv"))  # 3x168
    g["scene_in"] = read_csv_matrix(os.path.join(T, "sceneIn.csv"))
    g["object_noise_bound"] = np.array(0.0067364)
    g["object_expected_scale"] = np.array(0.955885)  # registration-test.cc:295, tol 1e-4
    g["object_expected_R"] = np.array([[0.9974, -0.0199, -0.0696], [0.0138, 0.9961, -0.0875],
                                       [0.0710, 0.0863, 0.9937]])
    g["object_expected_t"] = np.array([-0.1011, 0.0908, 0.1344])
    # bounds: scaled R<=0.25 t<=0.15 (:308-310), fixed-scale R<=0.2 t<=0.1 (:388-390)
    g["object_bounds"] = np.array([0.25, 0.15, 0.2, 0.1])

    # --- test/benchmark fixtures: registration-benchmark.cc:130-166 (loader), 276-374 (thresholds)
    thr = {
        1: [1e-5, 1e-5, 1e-5, 1e-5, 1e-5, 1e-5],
        2: [1e-5, 1e-5, 1e-5, 1e-5, 1e-5, 1e-5],
        3: [1e-5, 1e-5, 1e-5, 1e-5, 1e-5, 1e-5],
        4: [1e-5, 1e-5, 1e-5, 1e-5, 1e-5, 1e-5],
        5: [1e-5, 1e-5, 1e-5, 1e-5, 1e-5, 1e-5],
        6: [1e-2, 1e-2, 2e-2, 1e-5, 1e-3, 1e-3],  # s,R,t vs GT ; s,R,t vs MATLAB TEASER
    }
    for k in range(1, 7):
        d = os.path.join(B, "benchmark_%d" % k)
        g["bench%d_src" % k] = read_ply_xyz_f32(os.path.join(d, "src.ply"))
        g["bench%d_dst" % k] = read_ply_xyz_f32(os.path.join(d, "dst.ply"))
        with open(os.path.join(d, "parameters.txt")) as f:
            params = {}
            for line in f:
                if ":" in line:
                    a, b = line.split(":")
                    params[a.strip()] = float(b)
        g["bench%d_noise_bound" % k] = np.array(params["Noise Bound"])
        g["bench%d_outlier_ratio" % k] = np.array(params["Outlier Ratio"])
        for nm in ("R_ref", "R_est"):
            g["bench%d_%s" % (k, nm)] = read_csv_matrix(os.path.join(d, nm + ".csv"))
        for nm in ("t_ref", "t_est"):
            g["bench%d_%s" % (k, nm)] = read_csv_matrix(os.path.join(d, nm + ".csv")).reshape(3)
        for nm in ("s_ref", "s_est"):
            g["bench%d_%s" % (k, nm)] = np.array(float(open(os.path.join(d, nm + ".csv")).read()))
        g["bench%d_thresholds" % k] = np.array(thr[k])

    # --- clouds: Bunny (BASELINE config 1; examples/teaser_cpp_ply/teaser_cpp_ply.cc:44-68) and
    # the 1000-point smoke pair (registration-test.cc:21-105)
    g["bunny"] = read_ply_xyz_f32(os.path.join(T, "bun_zipper_res3.ply"))
    g["bunny_T"] = np.array([[9.96926560e-01, 6.68735757e-02, -4.06664421e-02, -1.15576939e-01],
                             [-6.61289946e-02, 9.97617877e-01, 1.94008687e-02, -3.87705398e-02],
                             [4.18675510e-02, -1.66517807e-02, 9.98977765e-01, 1.14874890e-01],
                             [0, 0, 0, 1]])  # teaser_cpp_ply.cc:62-68
    g["model1000"] = read_ply_xyz_f32(os.path.join(T, "1000point_model.ply"))
    g["scene1000"] = read_ply_xyz_f32(os.path.join(T, "1000point_scene.ply"))

    # --- toy graphs whose clique size the reference pins: test/teaser/graph-test.cc:131-305
    # stored as edge lists
    g["graph_k5_edges"] = np.array([(i, j) for i in range(5) for j in range(i + 1, 5)], dtype=np.int32)
    g["graph_4node_edges"] = np.array([(0, 2), (0, 3), (1, 2), (2, 3)], dtype=np.int32)  # :182-224 -> 3

    np.savez_compressed(args.out, **g)
    print("wrote", args.out, os.path.getsize(args.out), "bytes,", len(g), "arrays")


if __name__ == "__main__":
    main()
