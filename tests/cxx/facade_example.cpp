// Uses the drop-in C++ facade exactly like the reference's examples use the reference class
// (examples/teaser_cpp_ply/teaser_cpp_ply.cc:76-100): fill Params, solve(src, dst), read the
// solution and the inlier lists.  Exit code: 0 ok, 77 no MI355X visible (loud failure, no CPU path),
// 1 wrong result.
#include <cmath>
#include <cstdio>
#include <vector>

#include "teaser/registration.h"

int main() {
  const int n = 3000;
  std::vector<double> src((size_t)3 * n), dst((size_t)3 * n), Rt(9), tt(3);
  std::vector<unsigned char> inl((size_t)n);
  teaser_hip_synth_problem(20250523ull, n, 0.9, 0.01, src.data(), dst.data(), Rt.data(), tt.data(), inl.data());

  teaser::RobustRegistrationSolver::Params params;
  params.noise_bound = 0.01;
  params.cbar2 = 1;
  params.estimate_scaling = false;
  params.rotation_max_iterations = 100;
  params.rotation_gnc_factor = 1.4;
  params.rotation_estimation_algorithm =
      teaser::RobustRegistrationSolver::ROTATION_ESTIMATION_ALGORITHM::GNC_TLS;
  params.rotation_cost_threshold = 0.005;

  try {
    teaser::RobustRegistrationSolver solver(params);
    teaser::Matrix3X S(3, n), D(3, n);
    for (int i = 0; i < n; ++i)
      for (int r = 0; r < 3; ++r) {
        S(r, i) = src[(size_t)3 * i + r];
        D(r, i) = dst[(size_t)3 * i + r];
      }
    solver.solve(S, D);
    auto solution = solver.getSolution();
    double dR = 0, dt = 0;
    for (int r = 0; r < 3; ++r) {
      dt += std::pow(solution.translation(r) - tt[(size_t)r], 2);
      for (int c = 0; c < 3; ++c) dR += std::pow(solution.rotation(r, c) - Rt[(size_t)(3 * r + c)], 2);
    }
    const auto clique = solver.getInlierMaxClique();
    const auto tin = solver.getInputOrderedTranslationInliers();
    size_t planted = 0, hit = 0;
    for (int i = 0; i < n; ++i) planted += inl[(size_t)i];
    for (int v : clique) hit += inl[(size_t)v];
    std::printf("valid %d  |R-R*|_F %.3e  |t-t*| %.3e  clique %zu (planted %zu, hit %zu)  trans inliers %zu\n",
                (int)solution.valid, std::sqrt(dR), std::sqrt(dt), clique.size(), planted, hit, tin.size());
    // stage entry point (registration.h:593): rotation-only problem, dst = R* src exactly
    teaser::Matrix3X V1(3, 200), V2(3, 200);
    for (int i = 0; i < 200; ++i)
      for (int r = 0; r < 3; ++r) V1(r, i) = S(r, i) - S(r, 0);
    for (int i = 0; i < 200; ++i)
      for (int r = 0; r < 3; ++r)
        V2(r, i) = Rt[(size_t)(3 * r)] * V1(0, i) + Rt[(size_t)(3 * r + 1)] * V1(1, i) +
                   Rt[(size_t)(3 * r + 2)] * V1(2, i);
    const auto Rs = solver.solveForRotation(V1, V2);
    double dRs = 0;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) dRs += std::pow(Rs(r, c) - Rt[(size_t)(3 * r + c)], 2);
    std::printf("solveForRotation |R-R*|_F %.3e\n", std::sqrt(dRs));
    if (std::sqrt(dRs) > 1e-6) return 1;
    // second solve on the same object (the reference object is single-use; this one is not)
    solver.solve(S, D);
    const bool ok = solution.valid && std::sqrt(dR) < 0.02 && std::sqrt(dt) < 0.02 && hit >= planted &&
                    solver.getInlierMaxClique() == clique && !solver.getInlierGraph().empty();
    return ok ? 0 : 1;
  } catch (const std::runtime_error& e) {
    std::printf("facade: %s\n", e.what());
    return 77;
  }
}
