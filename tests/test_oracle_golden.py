"""Pin the CPU oracle against the reference's own golden vectors.

CPU-only.  Every expected value comes from tests/golden/teaser_golden.npz, which
tests/golden/make_golden.py collected from the reference's test fixtures / literals.
"""
import numpy as np
import pytest

from oracle import oracle
from util import angular_error, golden, is_clique

G = golden()


@pytest.mark.parametrize("case", [1, 2, 3])
def test_scalar_tls_known_answers(case):
    # reference test/teaser/tls-test.cc:25-84
    est, mask = oracle.scalar_tls(G["tls%d_x" % case], G["tls%d_r" % case])
    assert abs(est - float(G["tls%d_est" % case])) < float(G["tls_tol"])
    assert (mask == G["tls%d_mask" % case].astype(bool)).all()


def test_translation_known_answer():
    # reference test/teaser/translation-solver-test.cc:88-112
    t, mask = oracle.tls_translation(G["trans_v1"], G["trans_v2"], float(G["trans_noise_bound"]))
    assert np.linalg.norm(t - G["trans_expected_t"]) < float(G["trans_tol"])
    # survey probe reproduced the literal to 1e-14
    assert np.linalg.norm(t - G["trans_expected_t"]) < 1e-12


def test_translation_axis_cases():
    # translation-solver-test.cc:21-86: zero / unit-axis translations on a small cloud
    rng = np.random.default_rng(1)
    pts = rng.uniform(-1, 1, size=(3, 20))
    for a in range(3):
        d = pts.copy()
        d[a] += 1
        t, mask = oracle.tls_translation(pts, d, 0.01)
        e = np.zeros(3)
        e[a] = 1
        assert np.linalg.norm(t - e) < 1e-5 and mask.all()


def test_gnc_tls_rotation_known_answer():
    # reference test/teaser/rotation-solver-test.cc:221-250
    src = G["rot_src"].T  # 3x200
    R_exp = G["rot_expected_R"]
    dst = R_exp @ src
    mi, thr, fac, nb = G["rot_params"]
    out = oracle.gnc_tls_rotation(src, dst, nb, fac, int(mi), thr)
    assert angular_error(R_exp, out["R"]) < float(G["rot_tol"])
    assert np.linalg.norm(out["R"] - R_exp) < 1e-9
    assert out["inliers"].all()


def test_fgr_rotation_known_answer():
    # reference test/teaser/rotation-solver-test.cc:101-134 (Problem 3, one iteration)
    src = G["rot_src"].T
    R_exp = G["rot_expected_R"]
    dst = R_exp @ src
    mi, thr, fac, nb = G["fgr_params"]
    out = oracle.fgr_rotation(src, dst, nb, fac, int(mi), thr)
    assert angular_error(R_exp, out["R"]) < float(G["rot_tol"])
    assert out["iterations"] == 1 and out["inliers"].all()


def test_fgr_axis_rotations():
    # rotation-solver-test.cc:25-99: identity and axis rotations, Params{1000, 0.0337, 1.4, 1e-3}
    rng = np.random.default_rng(3)
    src = rng.uniform(-1, 1, size=(3, 10))
    th = 2.345
    c, s_ = np.cos(th), np.sin(th)
    for R in (np.eye(3), np.array([[1, 0, 0], [0, c, -s_], [0, s_, c]]),
              np.array([[c, 0, s_], [0, 1, 0], [-s_, 0, c]]), np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]])):
        out = oracle.fgr_rotation(src, R @ src, 1e-3, 1.4, 1000, 0.0337)
        assert angular_error(R, out["R"]) < 1e-5


def test_quatro_rotation_known_answer():
    # reference test/teaser/registration-test.cc:179-216: yaw-only rotation of the same cloud
    src = G["rot_src"].T
    R_exp = G["quatro_expected_R"]
    dst = R_exp @ src
    mi, thr, fac, nb = G["quatro_params"]
    out = oracle.quatro_rotation(src, dst, nb, fac, int(mi), thr)
    assert angular_error(R_exp, out["R"]) < 1e-5
    assert out["R"][2, 2] == 1 and (out["R"][2, :2] == 0).all() and (out["R"][:2, 2] == 0).all()
    # a general rotation: QUATRO returns the yaw that best aligns the xy-projections; the 2x2 block
    # is the polar (SVD) rotation of the projected correlation (utils.h:145-160)
    dst2 = G["rot_expected_R"] @ src
    out2 = oracle.quatro_rotation(src, dst2, nb, fac, int(mi), thr)
    w = np.ones(src.shape[1])
    H = (src[:2] * w) @ dst2[:2].T
    U, _, Vt = np.linalg.svd(H)
    V = Vt.T
    if np.linalg.det(U) * np.linalg.det(V) < 0:
        V[:, 1] *= -1
    if out2["iterations"] == 1:  # stopped at i = 0 (mu <= 0) or after one pass with w = 1
        assert np.linalg.norm(out2["R"][:2, :2] - V @ U.T) < 1e-9 or out2["cost"] < np.inf


def test_object_scene_end_to_end_fgr():
    # registration-test.cc:256-392 as written there: rotation_estimation_algorithm = FGR
    obj, scn = G["object_in"], G["scene_in"]
    nb = float(G["object_noise_bound"])
    bR1, bt1, bR2, bt2 = G["object_bounds"]
    o1 = oracle.solve(obj, scn, noise_bound=nb, estimate_scaling=1, rotation_cost_threshold=0.005,
                      rotation_estimation_algorithm=1)
    assert abs(o1["scale"] - float(G["object_expected_scale"])) < 1e-4
    assert angular_error(G["object_expected_R"], o1["rotation"]) <= bR1
    assert np.linalg.norm(o1["translation"] - G["object_expected_t"]) <= bt1
    o2 = oracle.solve(obj, scn, noise_bound=nb, estimate_scaling=0, rotation_cost_threshold=0.005,
                      rotation_estimation_algorithm=1)
    assert angular_error(G["object_expected_R"], o2["rotation"]) <= bR2
    assert np.linalg.norm(o2["translation"] - G["object_expected_t"]) <= bt2


def test_gnc_tls_axis_rotations():
    # rotation-solver-test.cc:137-219: identity and axis rotations, params {100,1e-12,1.4,1e-3}
    rng = np.random.default_rng(2)
    src = rng.uniform(-1, 1, size=(3, 10))
    th = 1.234
    c, s = np.cos(th), np.sin(th)
    Rs = [np.eye(3), np.array([[1, 0, 0], [0, c, -s], [0, s, c]]),
          np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]), np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])]
    for R in Rs:
        out = oracle.gnc_tls_rotation(src, R @ src, 1e-3, 1.4, 100, 1e-12)
        assert angular_error(R, out["R"]) < 1e-5


def test_svd_rot_against_lapack():
    # independent check of the oracle's own 3x3 Jacobi SVD (utils.h:121-136) with numpy/LAPACK
    rng = np.random.default_rng(3)
    for trial in range(50):
        k = int(rng.integers(3, 40))
        X = rng.normal(size=(3, k))
        R0 = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        if np.linalg.det(R0) < 0:
            R0[:, 0] *= -1
        Y = R0 @ X + 0.05 * rng.normal(size=(3, k))
        w = rng.uniform(0, 1, size=k)
        H = (X * w) @ Y.T
        U, S, Vt = np.linalg.svd(H)
        V = Vt.T
        if np.linalg.det(U) * np.linalg.det(V) < 0:
            V[:, 2] *= -1
        R_ref = V @ U.T
        R = oracle.svd_rot(X, Y, w)
        assert np.linalg.norm(R - R_ref) < 1e-10
        assert abs(np.linalg.det(R) - 1) < 1e-12


def test_scale_solvers_on_object_scene():
    # scale-solver-test.cc:23-130 + registration-test.cc:107-142 on objectIn/sceneIn (3x168)
    obj, scn = G["object_in"], G["scene_in"]
    nb = float(G["object_noise_bound"])
    a = oracle.compute_tims(obj)[0].T  # 3xM
    b = oracle.compute_tims(scn)[0].T
    s, mask = oracle.scale_inliers_mask(a, b, nb, 1.0, True)
    assert abs(s - float(G["object_expected_scale"])) < 1e-4
    # fixed-scale selector: all-in when dst == src; all-out when dst is far off scale
    s1, m1 = oracle.scale_inliers_mask(a, a, nb, 1.0, False)
    assert s1 == 1 and m1.all()
    s2, m2 = oracle.scale_inliers_mask(a, 10 * a, nb, 1.0, False)
    assert s2 == 1 and not m2.any()
    # one-out: perturb a single TIM
    c = a.copy()
    c[:, 5] *= 3
    s3, m3 = oracle.scale_inliers_mask(a, c, nb, 1.0, False)
    assert (~m3).sum() == 1 and not m3[5]
    # random-scale TLS (scale-solver-test.cc:46-69): exact data, scale recovered to 1e-5
    s4, m4 = oracle.scale_inliers_mask(a, 2.71 * a, nb, 1.0, True)
    assert abs(s4 - 2.71) < 1e-5 and m4.all()


def test_tim_pair_order():
    # registration.cc:531-547: k = i*N - i(i+1)/2 + (j-i-1), TIM = v_j - v_i, map = (i,j)
    rng = np.random.default_rng(4)
    v = rng.normal(size=(3, 7))
    tims, mp = oracle.compute_tims(v)
    k = 0
    for i in range(7):
        for j in range(i + 1, 7):
            assert (mp[k] == (i, j)).all()
            assert (tims[k] == v[:, j] - v[:, i]).all()
            k += 1


def test_toy_graph_cliques():
    # graph-test.cc:131-305: K5 -> 5 ({0..4}); 4-node graph -> 3; 4 isolated nodes -> 1
    bm = oracle.bitmap_from_edges(5, G["graph_k5_edges"])
    r = oracle.max_clique(bm, 5)
    assert list(r["clique"]) == [0, 1, 2, 3, 4] and r["unique"]
    bm = oracle.bitmap_from_edges(4, G["graph_4node_edges"])
    r = oracle.max_clique(bm, 4)
    assert list(r["clique"]) == [0, 2, 3] and r["unique"] and not r["exact_run"]
    bm = oracle.bitmap_from_edges(4, [])
    r = oracle.max_clique(bm, 4)
    assert len(r["clique"]) == 1 and not r["unique"]


def test_clique_against_networkx():
    nx = pytest.importorskip("networkx")
    rng = np.random.default_rng(5)
    for trial in range(25):
        n = int(rng.integers(5, 90))
        p = float(rng.uniform(0.1, 0.7))
        A = np.triu(rng.uniform(size=(n, n)) < p, 1)
        edges = np.argwhere(A)
        bm = oracle.bitmap_from_edges(n, edges)
        r = oracle.max_clique(bm, n)
        Gx = nx.Graph()
        Gx.add_nodes_from(range(n))
        Gx.add_edges_from(map(tuple, edges))
        cliques = list(nx.find_cliques(Gx))
        omega = max(len(c) for c in cliques)
        n_max = sum(1 for c in cliques if len(c) == omega)
        assert len(r["clique"]) == omega
        assert is_clique(A | A.T, r["clique"])
        assert r["unique"] == (n_max == 1)
        if n_max == 1:
            best = sorted([c for c in cliques if len(c) == omega][0])
            assert list(r["clique"]) == best


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6])
def test_benchmark_fixtures(k):
    # test/benchmark/registration-benchmark.cc:180-374: estimate_scaling=true, GNC-TLS thr 1e-12
    src = G["bench%d_src" % k].astype(np.float64).T
    dst = G["bench%d_dst" % k].astype(np.float64).T
    out = oracle.solve(src, dst, noise_bound=float(G["bench%d_noise_bound" % k]), cbar2=1.0,
                       estimate_scaling=1, rotation_max_iterations=100, rotation_gnc_factor=1.4,
                       rotation_cost_threshold=1e-12)
    assert out["valid"]
    sg, Rg, tg, se, Re, te = G["bench%d_thresholds" % k]
    assert abs(out["scale"] - float(G["bench%d_s_ref" % k])) <= sg
    assert angular_error(G["bench%d_R_ref" % k], out["rotation"]) <= Rg
    assert np.linalg.norm(out["translation"] - G["bench%d_t_ref" % k]) <= tg
    assert abs(out["scale"] - float(G["bench%d_s_est" % k])) <= se
    assert angular_error(G["bench%d_R_est" % k], out["rotation"]) <= Re
    assert np.linalg.norm(out["translation"] - G["bench%d_t_est" % k]) <= te


def test_object_scene_end_to_end():
    # registration-test.cc:256-392 (run with GNC-TLS; the reference's test uses FGR there, the
    # bounds are loose enough for either): scaled and fixed-scale problems
    obj, scn = G["object_in"], G["scene_in"]
    nb = float(G["object_noise_bound"])
    bR1, bt1, bR2, bt2 = G["object_bounds"]
    o1 = oracle.solve(obj, scn, noise_bound=nb, estimate_scaling=1, rotation_cost_threshold=0.005)
    assert abs(o1["scale"] - float(G["object_expected_scale"])) < 1e-4
    assert angular_error(G["object_expected_R"], o1["rotation"]) <= bR1
    assert np.linalg.norm(o1["translation"] - G["object_expected_t"]) <= bt1
    o2 = oracle.solve(obj, scn, noise_bound=nb, estimate_scaling=0, rotation_cost_threshold=0.005)
    assert o2["scale"] == 1
    assert angular_error(G["object_expected_R"], o2["rotation"]) <= bR2
    assert np.linalg.norm(o2["translation"] - G["object_expected_t"]) <= bt2
    # survey appendix C: hard clique instance (max_core+1 > omega): 34 / 34
    assert len(o1["max_clique"]) == 34 and o1["max_core"] == 39 and o1["clique_exact_run"]
    assert len(o2["max_clique"]) == 34 and o2["max_core"] == 36 and o2["clique_exact_run"]
    assert o1["num_edges"] == 2087 and o2["num_edges"] == 2008


def test_outlier_detection_structure():
    # registration-test.cc:394-467: N=20, far outliers -> max clique == exact inlier index set
    rng = np.random.default_rng(7)
    for n_out in range(1, 6):
        src = rng.uniform(-1, 1, size=(3, 20))
        R = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        if np.linalg.det(R) < 0:
            R[:, 0] *= -1
        t = rng.uniform(-1, 1, size=(3, 1))
        dst = R @ src + t
        out_idx = rng.choice(20, size=n_out, replace=False)
        dst[:, out_idx] += rng.uniform(5, 10, size=(3, n_out))
        o = oracle.solve(src, dst, noise_bound=0.01, estimate_scaling=0, rotation_cost_threshold=1e-12)
        expect = sorted(set(range(20)) - set(out_idx.tolist()))
        assert list(o["max_clique"]) == expect and o["clique_unique"]
        assert angular_error(R, o["rotation"]) < 1e-6
        assert np.linalg.norm(o["translation"] - t.ravel()) < 1e-6


def test_thousand_point_smoke():
    # registration-test.cc:21-105 (LargeModel*): smoke, no asserts on the result in the reference
    src = G["model1000"].astype(np.float64).T
    dst = G["scene1000"].astype(np.float64).T
    o = oracle.solve(src, dst, noise_bound=0.0067364, estimate_scaling=0, rotation_cost_threshold=0.005)
    assert o["valid"] and len(o["max_clique"]) > 2


def test_python_mirror_names_and_tim_helper():
    """Host-side pieces of the Python mirror that need no GPU: the drop-in module name, the enum
    and Params surface of python/teaserpp_python/teaserpp_python.cc:27-110, and the lazily rebuilt
    TIM product (pair order of registration.cc:531) against the oracle's computeTIMs."""
    import teaserpp_python as t
    p = t.RobustRegistrationSolver.Params()
    assert (p.noise_bound, p.cbar2, p.estimate_scaling, p.rotation_gnc_factor, p.rotation_max_iterations,
            p.rotation_cost_threshold, p.kcore_heuristic_threshold, p.max_clique_time_limit) == \
        (0.01, 1, True, 1.4, 100, 1e-6, 0.5, 3600)
    assert int(t.RotationEstimationAlgorithm.QUATRO) == 2 and int(t.InlierSelectionMode.NONE) == 3
    assert int(t.InlierGraphFormulation.COMPLETE) == 1 and t.OMP_MAX_THREADS >= 1
    assert t.RobustRegistrationSolver.ROTATION_ESTIMATION_ALGORITHM is t.RotationEstimationAlgorithm
    # the certifier names of teaserpp_python.cc:71-74, 249-291 (defaults: certification.h:71-104)
    cp = t.DRSCertifier.Params()
    assert (cp.noise_bound, cp.cbar2, cp.sub_optimality, cp.max_iterations, cp.gamma_tau) == \
        (0.01, 1, 1e-3, 2e2, 1.999999)
    assert cp.eig_decomposition_solver == t.EigSolverType.EIGEN and int(t.EigSolverType.SPECTRA) == 1
    assert "is_optimal=False" in repr(t.CertificationResult())
    rng = np.random.default_rng(5)
    v = rng.normal(size=(3, 37))
    tims, mp = oracle.compute_tims(v)
    mine = t.RobustRegistrationSolver._compute_tims(np.ascontiguousarray(v.T))
    assert np.array_equal(mine, tims.T if tims.shape[0] != 3 else tims)


def test_config_fixture_is_what_the_oracle_returns():
    """tests/golden/config_golden.json (tests/golden/make_config_golden.py) holds oracle results for
    the full-size BASELINE configs; the small entries are re-solved here so that a drift of the oracle
    (flags, rebuild) against the committed fixture is caught on the CPU, and the materialising
    (reference-faithful) front half must give the same graph as the streaming one."""
    import hashlib
    import importlib
    import json
    import os
    from util import ROOT
    tp = importlib.import_module("teaser-plusplus_amd")
    fx_all = json.load(open(os.path.join(ROOT, "tests", "golden", "config_golden.json")))
    kw = dict(noise_bound=0.01, cbar2=1.0, estimate_scaling=0, rotation_gnc_factor=1.4,
              rotation_max_iterations=100, rotation_cost_threshold=0.005)
    for name in ("config4_b0", "config4_b127"):
        fx = fx_all[name]
        pr = tp.synth_problem(fx["seed"], fx["n"], fx["outlier_ratio"], fx["noise_bound"])
        for materialise in (False, True):
            o = oracle.solve(pr["src"], pr["dst"], materialise=materialise, **kw)
            assert o["max_clique"].tolist() == fx["max_clique"] and o["num_edges"] == fx["num_edges"]
            assert o["rotation_inliers"].tolist() == fx["rotation_inliers"]
            assert o["translation_inliers"].tolist() == fx["translation_inliers"]
            assert np.abs(o["rotation"].reshape(-1) - np.array(fx["rotation"])).max() < 1e-12
            assert np.abs(o["translation"] - np.array(fx["translation"])).max() < 1e-12
        _, bm = oracle.inlier_bitmap(pr["src"], pr["dst"], 0.01, 1.0, False)
        assert hashlib.sha256(np.ascontiguousarray(bm).tobytes()).hexdigest() == fx["bitmap_sha256"]
    assert fx_all["config3"]["n"] == 50000 and fx_all["config3"]["clique_unique"]


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_reference_no_max_clique_and_clique_finder_modes(seed):
    """registration-test.cc:469-533 (use_max_clique = false) and :535-680 (PMC_EXACT, PMC_HEU, KCORE_HEU, NONE) on the
    reference's own scenario (tests/reference_cases.py), at the reference's thresholds: rotation within 0.2 rad,
    translation within 0.1."""
    from reference_cases import PARAMS, T_REF, angular_error, case
    src, tgt, outliers = case(seed)
    runs = [dict(PARAMS, use_max_clique=0)] + [dict(PARAMS, inlier_selection_mode=m) for m in (0, 1, 2, 3)]
    for kw in runs:
        o = oracle.solve(src, tgt, **dict(kw, estimate_scaling=0))
        assert o["valid"]
        assert angular_error(T_REF[:, :3], np.asarray(o["rotation"]).reshape(3, 3)) <= 0.2
        assert np.linalg.norm(T_REF[:, 3] - np.asarray(o["translation"]).ravel()) <= 0.1
        if kw.get("use_max_clique", 1) == 0 or kw.get("inlier_selection_mode") == 3:
            assert len(o["max_clique"]) == src.shape[1]     # no inlier selection: every correspondence goes on
        elif kw.get("inlier_selection_mode") in (0, 1):
            assert o["max_clique"].tolist() == np.flatnonzero(~outliers).tolist()
