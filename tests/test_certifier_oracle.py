"""The certifier oracle (oracle/certifier.py, a numpy restatement of teaser/src/certification.cc) against the
reference's own fixtures (tests/golden/certifier_golden.npz <- test/teaser/data/certification_*_instances):
the checks of test/teaser/certification-test.cc:355-525, at its tolerance (ACCEPTABLE_ERROR = 1e-7; matrices
with Eigen's isApprox: ||a - b|| <= 1e-12 * min(||a||, ||b||) is far tighter than the CSV precision allows, so
1e-9 relative is used for them)."""
import os

import numpy as np
import pytest

from oracle import certifier as C

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "certifier_golden.npz"))
TOL = 1e-7
SMALL = (1, 2, 3)


def close(a, b, rel=1e-9):
    return np.linalg.norm(a - b) <= rel * max(1.0, min(np.linalg.norm(a), np.linalg.norm(b)))


def case(kind, c):
    g = lambda n: G["%s%d_%s" % (kind, c, n)]
    nb, cbar2, iters = g("params")
    q = g("q_est").reshape(-1)  # stored as x y z w (certification-test.cc:163-166)
    return dict(R=g("R_est"), q=q, theta=g("theta_est").reshape(-1), v1=g("v1"), v2=g("v2"), nb=float(nb),
                cbar2=float(cbar2), iters=int(iters), g=g)


@pytest.mark.parametrize("c", SMALL)
def test_omega_and_block_diag(c):
    d = case("small", c)
    assert close(C.omega1(d["q"]), d["g"]("omega"))
    npm = 4 * (d["v1"].shape[1] + 1)
    assert close(C.block_diag_omega(npm, d["q"]), d["g"]("block_diag_omega"))
    # the quaternion the certifier derives from R_est is the fixture's q_est (up to sign)
    q = C.rotation_to_quaternion(d["R"])
    assert min(np.linalg.norm(q - d["q"]), np.linalg.norm(q + d["q"])) < 1e-7


@pytest.mark.parametrize("c", SMALL)
def test_q_cost(c):
    d = case("small", c)
    assert close(C.q_cost(d["v1"], d["v2"], d["nb"], d["cbar2"]), d["g"]("Q_cost"))


@pytest.mark.parametrize("c", SMALL)
def test_lambda_guess(c):
    d = case("small", c)
    L = C.lambda_guess(d["R"], d["theta"], d["v1"], d["v2"], d["nb"], d["cbar2"])
    assert close(L, d["g"]("lambda_bar_init"), rel=1e-8)


@pytest.mark.parametrize("c", SMALL)
def test_linear_projection(c):
    d = case("small", c)
    A = C.linear_projection(np.concatenate([[1.0], d["theta"]]))
    assert close(A, d["g"]("A_inv"))


@pytest.mark.parametrize("c", SMALL)
def test_optimal_dual_projection(c):
    d = case("small", c)
    thp = np.concatenate([[1.0], d["theta"]])
    Wd = C.optimal_dual_projection(d["g"]("W_1st_iter"), thp, d["g"]("A_inv"))
    assert np.abs(Wd - d["g"]("W_dual_1st_iter")).max() < TOL


@pytest.mark.parametrize("c", SMALL)
def test_suboptimality_gap_first_iteration(c):
    d = case("small", c)
    gap = C.suboptimality_gap(d["g"]("M_affine_1st_iter"), float(d["g"]("mu")[0, 0]), d["v1"].shape[1])
    assert abs(gap - float(d["g"]("suboptimality_1st_iter")[0, 0])) < TOL


@pytest.mark.parametrize("kind,c", [("small", 1), ("small", 2), ("small", 3), ("large", 1), ("large", 2)])
def test_certify_trajectory(kind, c):
    d = case(kind, c)
    want = d["g"]("suboptimality_traj").reshape(-1)
    out = C.certify(d["R"], d["v1"], d["v2"], d["theta"], noise_bound=d["nb"], cbar2=d["cbar2"],
                    max_iterations=d["iters"], return_first_iteration=(kind == "small"))
    got = out["suboptimality_traj"]
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.abs(got - want).max() < TOL
    assert abs(out["best_suboptimality"] - want.min()) < TOL
    if kind == "small":
        # (W_1st_iter / M_affine_1st_iter of the fixtures are inputs of the per-function tests above, produced
        # by the MATLAB original from another starting point; mu is the same quantity)
        f = out["first_iteration"]
        assert abs(f["mu"] - float(d["g"]("mu")[0, 0])) < 1e-7 * max(1.0, abs(f["mu"]))
