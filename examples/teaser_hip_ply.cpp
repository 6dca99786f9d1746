// Registration of a PLY cloud against a transformed, noisy, outlier-ridden copy of itself on the
// MI355X, through the drop-in C++ facade only (teaser/ply_io.h, teaser/registration.h) -- the
// workflow of the reference's examples/teaser_cpp_ply/teaser_cpp_ply.cc (BASELINE config 1: Bunny,
// 1889 correspondences), with a seeded generator instead of std::random_device.
//
//   g++ -std=c++17 -O2 -Iinclude examples/teaser_hip_ply.cpp -o teaser_hip_ply
//       -Lteaser-plusplus_amd -lteaser_hip -Wl,-rpath,$PWD/teaser-plusplus_amd -Wl,-rpath,/opt/rocm/lib
//   ./teaser_hip_ply bun_zipper_res3.ply
// Exit code: 0 registration within 0.01 rad / 1e-3 m of the applied transform, 1 otherwise,
// 77 no MI355X visible, 2 usage / unreadable PLY.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <stdexcept>

#include "teaser/ply_io.h"
#include "teaser/registration.h"

namespace {
constexpr double kNoiseBound = 0.001;  // NOISE_BOUND of the reference example
constexpr int kOutlierDraws = 1700;    // N_OUTLIERS of the reference example (drawn with replacement)

struct Rng {  // splitmix64
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};
}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) {
    std::printf("usage: %s <cloud.ply>\n", argv[0]);
    return 2;
  }
  teaser::PLYReader reader;
  teaser::PointCloud src_cloud;
  if (reader.read(argv[1], src_cloud) != 0 || src_cloud.size() < 3) {
    std::printf("could not read %s\n", argv[1]);
    return 2;
  }
  const int64_t N = (int64_t)src_cloud.size();

  // the SE(3) transform applied by the reference example (teaser_cpp_ply.cc:62-68)
  const double R[3][3] = {{9.96926560e-01, 6.68735757e-02, -4.06664421e-02},
                          {-6.61289946e-02, 9.97617877e-01, 1.94008687e-02},
                          {4.18675510e-02, -1.66517807e-02, 9.98977765e-01}};
  const double t[3] = {-1.15576939e-01, -3.87705398e-02, 1.14874890e-01};

  teaser::Matrix3X src(3, N), tgt(3, N);
  for (int64_t i = 0; i < N; ++i) {
    const double p[3] = {src_cloud[(size_t)i].x, src_cloud[(size_t)i].y, src_cloud[(size_t)i].z};
    for (int r = 0; r < 3; ++r) {
      src(r, i) = p[r];
      tgt(r, i) = R[r][0] * p[0] + R[r][1] * p[1] + R[r][2] * p[2] + t[r];
    }
  }
  // noise: uniform in +-NOISE_BOUND/2 per axis; outliers: 1700 draws, each shifted by (k, k, k), k in 5..10
  Rng rng{20250523ull};
  for (int64_t i = 0; i < N; ++i)
    for (int r = 0; r < 3; ++r) tgt(r, i) += (2 * rng.uniform() - 1) * kNoiseBound / 2;
  for (int k = 0; k < kOutlierDraws; ++k) {
    const int64_t c = (int64_t)(rng.next() % (uint64_t)N);
    const double shift = 5 + (double)(rng.next() % 6);
    for (int r = 0; r < 3; ++r) tgt(r, c) += shift;
  }

  teaser::RobustRegistrationSolver::Params params;  // teaser_cpp_ply.cc:79-87
  params.noise_bound = kNoiseBound;
  params.cbar2 = 1;
  params.estimate_scaling = false;
  params.rotation_max_iterations = 100;
  params.rotation_gnc_factor = 1.4;
  params.rotation_estimation_algorithm = teaser::RobustRegistrationSolver::ROTATION_ESTIMATION_ALGORITHM::GNC_TLS;
  params.rotation_cost_threshold = 0.005;

  try {
    teaser::RobustRegistrationSolver solver(params);
    solver.solve(src, tgt);  // first call sizes the device arenas
    const auto begin = std::chrono::steady_clock::now();
    solver.solve(src, tgt);
    const auto end = std::chrono::steady_clock::now();
    const auto solution = solver.getSolution();

    double tr = 0, dt = 0;
    for (int r = 0; r < 3; ++r) {
      dt += std::pow(solution.translation(r) - t[r], 2);
      for (int c = 0; c < 3; ++c) tr += R[c][r] * solution.rotation(c, r);  // trace(R^T R_est)
    }
    const double ang = std::fabs(std::acos(std::fmin(1.0, std::fmax(-1.0, (tr - 1) / 2))));
    std::printf("correspondences %lld  max clique %zu  rotation inliers %zu  translation inliers %zu\n",
                (long long)N, solver.getInlierMaxClique().size(), solver.getRotationInliers().size(),
                solver.getTranslationInliers().size());
    std::printf("rotation error %.3e rad  translation error %.3e m  solve %.3f ms\n", ang, std::sqrt(dt),
                std::chrono::duration<double, std::milli>(end - begin).count());
    return (solution.valid && ang < 0.01 && std::sqrt(dt) < 1e-3) ? 0 : 1;
  } catch (const std::runtime_error& e) {
    std::printf("%s\n", e.what());
    return 77;
  }
}
