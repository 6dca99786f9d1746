"""Drop-in module name for users of the reference's Python binding: `import teaserpp_python`
resolves to the MI355X implementation (teaser-plusplus_amd), which mirrors the names registered in
python/teaserpp_python/teaserpp_python.cc:27-177 (RobustRegistrationSolver, its Params, the three
enums, RegistrationSolution, OMP_MAX_THREADS) and :71-74, 249-291 (DRSCertifier, its Params,
CertificationResult, EigSolverType)."""
import importlib as _importlib

_impl = _importlib.import_module("teaser-plusplus_amd")

RobustRegistrationSolver = _impl.RobustRegistrationSolver
RegistrationSolution = _impl.RegistrationSolution
RotationEstimationAlgorithm = _impl.RotationEstimationAlgorithm
InlierGraphFormulation = _impl.InlierGraphFormulation
InlierSelectionMode = _impl.InlierSelectionMode
OMP_MAX_THREADS = _impl.OMP_MAX_THREADS
DRSCertifier = _impl.DRSCertifier
CertificationResult = _impl.CertificationResult
EigSolverType = _impl.EigSolverType

__all__ = ["RobustRegistrationSolver", "RegistrationSolution", "RotationEstimationAlgorithm",
           "InlierGraphFormulation", "InlierSelectionMode", "OMP_MAX_THREADS", "DRSCertifier",
           "CertificationResult", "EigSolverType"]
