"""Inputs of two of the reference's own registration tests, restated (test/teaser/registration-test.cc:469-533 NoMaxClique,
:535-680 CliqueFinderModes): a random 20-point cloud (Eigen's Random(): uniform in [-1, 1]), the fixed transformation T of
the test, 1-5 correspondences turned into outliers by a translation of 5 .. 10 along all axes.  The reference draws them
from std::random_device; here the draws are seeded (several seeds)."""
import numpy as np

T_REF = np.array([[9.96926560e-01, 6.68735757e-02, -4.06664421e-02, -1.15576939e-01],
                  [-6.61289946e-02, 9.97617877e-01, 1.94008687e-02, -3.87705398e-02],
                  [4.18675510e-02, -1.66517807e-02, 9.98977765e-01, 1.14874890e-01]])
PARAMS = dict(noise_bound=0.01, cbar2=1.0, estimate_scaling=False, rotation_max_iterations=100, rotation_gnc_factor=1.4,
              rotation_cost_threshold=0.005)  # rotation_estimation_algorithm = GNC_TLS (the default)


def case(seed, n=20):
    rng = np.random.default_rng(seed)
    src = rng.uniform(-1.0, 1.0, size=(3, n))
    tgt = T_REF[:, :3] @ src + T_REF[:, 3:4]
    outliers = np.zeros(n, dtype=bool)
    for _ in range(int(rng.integers(1, 6))):          # dis1(1, 5)
        c = int(rng.integers(0, n))                    # dis2(0, N - 1)
        outliers[c] = True
        tgt[:, c] += float(rng.integers(5, 11))        # dis3(5, 10), added to x, y and z
    return src, tgt, outliers


def angular_error(Ra, Rb):
    """teaser::test::getAngularError (test/test-tools/test_utils.h): acos of the clamped (trace - 1) / 2"""
    return float(np.arccos(np.clip((np.trace(Ra.T @ Rb) - 1.0) / 2.0, -1.0, 1.0)))
