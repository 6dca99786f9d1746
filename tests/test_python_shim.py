"""`import teaserpp_python` is a drop-in for the reference's module: the names of
python/teaserpp_python/__init__.py:4-57 and both constructors of teaserpp_python.cc:82-103 (no GPU needed:
constructor-argument handling is checked up to the point a device would be opened).  When the reference tree is
present (this container, not the GPU box), every attribute its own Python examples read from the module and
from the solver object is checked to exist."""
import ast
import importlib
import os

import pytest

import teaserpp_python as t

tp = importlib.import_module("teaser-plusplus_amd")
S = t.RobustRegistrationSolver


def test_module_surface_matches_reference_init():
    for name in ("OMP_MAX_THREADS", "CertificationResult", "DRSCertifier", "EigSolverType",
                 "InlierGraphFormulation", "InlierSelectionMode", "RegistrationSolution",
                 "RobustRegistrationSolver", "RotationEstimationAlgorithm", "RobustRegistrationSolverParams"):
        assert hasattr(t, name), name
    # v1.0 aliases (python/teaserpp_python/__init__.py:16-20)
    assert S.ROTATION_ESTIMATION_ALGORITHM is t.RotationEstimationAlgorithm
    assert S.INLIER_SELECTION_MODE is t.InlierSelectionMode
    assert S.INLIER_GRAPH_FORMULATION is t.InlierGraphFormulation
    assert t.DRSCertifier.EIG_SOLVER_TYPE is t.EigSolverType
    assert isinstance(S.params, property)


def test_params_namedtuple_fields_and_defaults():
    p = t.RobustRegistrationSolverParams()
    assert p._fields == S._CTOR_ARGS
    assert p == (0.01, 1, True, t.RotationEstimationAlgorithm.GNC_TLS, 1.4, 100, 1e-6,
                 t.InlierGraphFormulation.CHAIN, t.InlierSelectionMode.PMC_EXACT, 0.5, True, True, 3000,
                 t.OMP_MAX_THREADS)
    # registration.h:419-514: the struct's own defaults are the same values, except the time limit
    # (3600 in the struct, registration.h:508; 3000 in the positional constructor, teaserpp_python.cc:102)
    d = S.Params()
    for k in p._fields:
        if k == "max_clique_time_limit":
            assert (getattr(d, k), getattr(p, k)) == (3600, 3000)
        else:
            assert getattr(d, k) == getattr(p, k), k


def test_positional_constructor_maps_by_position_not_to_device():
    P = S._params_from_ctor_args((0.05, 2.0, False), {})
    assert (P.noise_bound, P.cbar2, P.estimate_scaling) == (0.05, 2.0, False)
    P = S._params_from_ctor_args(tuple(t.RobustRegistrationSolverParams(
        cbar2=1, noise_bound=1, estimate_scaling=True,
        rotation_estimation_algorithm=t.RotationEstimationAlgorithm.GNC_TLS, rotation_gnc_factor=1.4,
        rotation_max_iterations=100, rotation_cost_threshold=1e-12)), {})  # teaserpp_example.py:25-35
    assert P.noise_bound == 1 and P.rotation_cost_threshold == 1e-12 and P.max_clique_time_limit == 3000
    c = P.to_c()
    assert c.noise_bound == 1.0 and c.estimate_scaling == 1 and c.rotation_max_iterations == 100
    P = S._params_from_ctor_args((), dict(noise_bound=0.2, inlier_selection_mode=t.InlierSelectionMode.PMC_HEU))
    assert P.noise_bound == 0.2 and P.inlier_selection_mode == 1 and P.estimate_scaling is True
    P = S._params_from_ctor_args((), {})
    assert P.noise_bound == 0.01
    q = S.Params()
    q.noise_bound = 0.3
    assert S._params_from_ctor_args((q,), {}) is q
    assert S._params_from_ctor_args((), {"params": q}) is q
    with pytest.raises(TypeError):
        S._params_from_ctor_args((0.1,), {"noise_bound": 0.2})
    with pytest.raises(TypeError):
        S._params_from_ctor_args((), {"nonsense": 1})
    with pytest.raises(TypeError):
        S._params_from_ctor_args(tuple(range(15)), {})
    with pytest.raises(TypeError):
        S._params_from_ctor_args((q,), {"cbar2": 2})


def test_positional_constructor_reaches_the_device_open():
    """`RobustRegistrationSolver(*params)` gets past argument handling: without a GPU the only error is
    NO_DEVICE (and with one the call succeeds)."""
    if tp.device_count() > 0:
        s = S(*t.RobustRegistrationSolverParams(noise_bound=0.5))
        assert s.params.noise_bound == 0.5
        return
    with pytest.raises(tp.TeaserHipError) as e:
        S(*t.RobustRegistrationSolverParams(noise_bound=0.5))
    assert e.value.status == 3


REF_EXAMPLES = ["/root/reference/python/teaserpp_python/teaserpp_example.py",
                "/root/reference/examples/teaser_python_ply/teaser_python_ply.py",
                "/root/reference/examples/teaser_python_fpfh_icp/example.py",
                "/root/reference/examples/teaser_python_fpfh_icp/helpers.py",
                "/root/reference/examples/teaser_python_3dsmooth/teaser_python_3dsmooth.py"]


def _attribute_chains(tree):
    """Dotted names read in the file: 'teaserpp_python.X.Y' and 'solver.attr' style chains."""
    out = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute):
            parts, cur = [], node
            while isinstance(cur, ast.Attribute):
                parts.append(cur.attr)
                cur = cur.value
            if isinstance(cur, ast.Name):
                out.add((cur.id,) + tuple(reversed(parts)))
    return out


@pytest.mark.parametrize("path", REF_EXAMPLES)
def test_reference_python_examples_find_every_name_they_use(path):
    if not os.path.exists(path):
        pytest.skip("reference tree not present (GPU box)")
    tree = ast.parse(open(path).read())
    solver_names = {"solver", "teaser_solver"}
    seen = 0
    for chain in _attribute_chains(tree):
        if chain[0] == "teaserpp_python":
            obj = t
            for a in chain[1:]:
                assert hasattr(obj, a), "teaserpp_python.%s missing (%s)" % (".".join(chain[1:]), path)
                obj = getattr(obj, a)
            seen += 1
        elif chain[0] in solver_names:
            assert hasattr(S, chain[1]), "RobustRegistrationSolver.%s missing (%s)" % (chain[1], path)
            seen += 1
        elif chain[0] == "solver_params":
            assert hasattr(S.Params(), chain[1]), "Params.%s missing (%s)" % (chain[1], path)
            seen += 1
    assert seen > 0
