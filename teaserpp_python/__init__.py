"""Drop-in module name for users of the reference's Python binding: `import teaserpp_python`
resolves to the MI355X implementation (teaser-plusplus_amd), which mirrors the names registered in
python/teaserpp_python/teaserpp_python.cc:27-177 (RobustRegistrationSolver with both constructors,
its Params, the three enums, RegistrationSolution, OMP_MAX_THREADS), :71-74, 249-291 (DRSCertifier,
its Params, CertificationResult, EigSolverType) and the pure-Python layer on top of them,
python/teaserpp_python/__init__.py:16-57 (the v1.0 enum aliases, RobustRegistrationSolverParams and
the `params` property), so that python/teaserpp_python/teaserpp_example.py and
examples/teaser_python_ply/teaser_python_ply.py's solver sequence run unmodified
(tests/test_gpu_python_shim.py executes both)."""
import importlib as _importlib

_impl = _importlib.import_module("teaser-plusplus_amd")

RobustRegistrationSolver = _impl.RobustRegistrationSolver
RobustRegistrationSolverParams = _impl.RobustRegistrationSolverParams
RegistrationSolution = _impl.RegistrationSolution
RotationEstimationAlgorithm = _impl.RotationEstimationAlgorithm
InlierGraphFormulation = _impl.InlierGraphFormulation
InlierSelectionMode = _impl.InlierSelectionMode
OMP_MAX_THREADS = _impl.OMP_MAX_THREADS
DRSCertifier = _impl.DRSCertifier
CertificationResult = _impl.CertificationResult
EigSolverType = _impl.EigSolverType

# Backwards compatibility with v1.0 (python/teaserpp_python/__init__.py:16-20); the implementation
# classes carry the same aliases, restated here so the shim does not depend on that.
RobustRegistrationSolver.ROTATION_ESTIMATION_ALGORITHM = RotationEstimationAlgorithm
RobustRegistrationSolver.INLIER_SELECTION_MODE = InlierSelectionMode
RobustRegistrationSolver.INLIER_GRAPH_FORMULATION = InlierGraphFormulation
DRSCertifier.EIG_SOLVER_TYPE = EigSolverType

__all__ = ["RobustRegistrationSolver", "RobustRegistrationSolverParams", "RegistrationSolution",
           "RotationEstimationAlgorithm", "InlierGraphFormulation", "InlierSelectionMode",
           "OMP_MAX_THREADS", "DRSCertifier", "CertificationResult", "EigSolverType"]
