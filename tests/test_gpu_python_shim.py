"""The call sequences of the reference's own Python examples, against `import teaserpp_python` on the GPU:
python/teaserpp_python/teaserpp_example.py (NamedTuple params -> `RobustRegistrationSolver(*params)` -> the
`solution`, `scale_inliers`, `translation_inliers`, `translation_inliers_map` properties) and
examples/teaser_python_ply/teaser_python_ply.py:52-66 (`RobustRegistrationSolver.Params()` filled field by
field, v1.0 enum alias, `solve`, `getSolution`).  The reference files themselves are executed unmodified with
runpy where the reference tree exists (not on the GPU box); the sequences below are restatements with the results
checked (the examples only print)."""
import os
import runpy

import numpy as np
import pytest

import teaserpp_python
from util import angular_error

pytestmark = pytest.mark.gpu

ROTATION = np.array([[0.98370992, 0.17903344, -0.01618098],
                     [-0.04165862, 0.13947877, -0.98934839],
                     [-0.17486954, 0.9739059, 0.14466493]])  # teaserpp_example.py:15-17


def test_example_sequence_namedtuple_and_properties():
    rng = np.random.default_rng(7)
    src = rng.random((3, 20))
    scale, translation = 1.5, np.array([[1.0], [0.0], [-1.0]])
    dst = scale * (ROTATION @ src) + translation
    dst[:, 1] += 10
    dst[:, 9] += 15
    params = teaserpp_python.RobustRegistrationSolverParams(
        cbar2=1, noise_bound=1, estimate_scaling=True,
        rotation_estimation_algorithm=teaserpp_python.RotationEstimationAlgorithm.GNC_TLS,
        rotation_gnc_factor=1.4, rotation_max_iterations=100, rotation_cost_threshold=1e-12)
    solver = teaserpp_python.RobustRegistrationSolver(*params)
    assert solver.params == params and isinstance(solver.params, teaserpp_python.RobustRegistrationSolverParams)
    solver.solve(src, dst)
    sol = solver.solution
    assert sol.valid
    assert abs(sol.scale - scale) < 1e-6
    assert angular_error(ROTATION, sol.rotation) < 1e-6
    assert np.linalg.norm(sol.translation - translation.ravel()) < 1e-6
    assert "scale" in repr(sol)
    pairs = solver.scale_inliers
    assert len(pairs) == len(solver.getScaleInliers()) > 0
    assert all(1 not in p and 9 not in p for p in pairs)  # "they should not include the outlier points"
    tmap = np.asarray(solver.translation_inliers_map).ravel()
    tin = solver.translation_inliers
    assert sorted(tmap[tin].tolist()) == [i for i in range(20) if i not in (1, 9)]


def test_ply_example_construction_sequence():
    rng = np.random.default_rng(11)
    N, NOISE_BOUND = 400, 0.05
    src = rng.random((3, N))
    T = np.array([[9.96926560e-01, 6.68735757e-02, -4.06664421e-02, -1.15576939e-01],
                  [-6.61289946e-02, 9.97617877e-01, 1.94008687e-02, -3.87705398e-02],
                  [4.18675510e-02, -1.66517807e-02, 9.98977765e-01, 1.14874890e-01],
                  [0, 0, 0, 1]])  # teaser_python_ply.py:32-36
    dst = T[:3, :3] @ src + T[:3, 3:4]
    dst += (rng.random((3, N)) - 0.5) * 2 * NOISE_BOUND / np.sqrt(3)
    out = rng.integers(0, N, size=300)
    for i in out:
        dst[:, i] += 5 + rng.random(3) * 5
    solver_params = teaserpp_python.RobustRegistrationSolver.Params()
    solver_params.cbar2 = 1
    solver_params.noise_bound = NOISE_BOUND
    solver_params.estimate_scaling = False
    solver_params.rotation_estimation_algorithm = \
        teaserpp_python.RobustRegistrationSolver.ROTATION_ESTIMATION_ALGORITHM.GNC_TLS
    solver_params.rotation_gnc_factor = 1.4
    solver_params.rotation_max_iterations = 100
    solver_params.rotation_cost_threshold = 1e-12
    solver = teaserpp_python.RobustRegistrationSolver(solver_params)
    solver.solve(src, dst)
    solution = solver.getSolution()
    assert solution.valid and solution.scale == 1.0
    assert angular_error(T[:3, :3], solution.rotation) < 0.1
    assert np.linalg.norm(T[:3, 3] - solution.translation) < 0.1
    assert solver.getParams() is solver_params
    assert solver.params.noise_bound == NOISE_BOUND and solver.params.estimate_scaling is False


def test_keyword_constructor_and_device_keyword():
    solver = teaserpp_python.RobustRegistrationSolver(noise_bound=0.02, estimate_scaling=False, device=0)
    assert solver.params.noise_bound == 0.02 and solver.params.rotation_cost_threshold == 1e-6
    solver = teaserpp_python.RobustRegistrationSolver(0.03, 1, False)  # positional: never a device index
    assert solver.params[:3] == (0.03, 1, False)


def test_reference_example_file_runs_unmodified(capsys):
    path = "/root/reference/python/teaserpp_python/teaserpp_example.py"
    if not os.path.exists(path):
        pytest.skip("reference tree not present on the GPU box")
    runpy.run_path(path, run_name="__main__")
    assert "Translation inliers map is:" in capsys.readouterr().out
