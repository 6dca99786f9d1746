// The reference's examples/teaser_cpp_fpfh/teaser_cpp_fpfh.cc:40-110 workflow through the drop-in headers:
// cloud + transformed noisy copy -> teaser::FPFHEstimation (0.02, 0.04) -> teaser::Matcher (cross check) ->
// RobustRegistrationSolver::solve(cloud, cloud, correspondences).  argv[1]: ascii PLY (x y z vertices).
// Exit code: 0 ok, 77 no MI355X visible, 2 unreadable input, 1 wrong result.
#include <cmath>
#include <cstdio>
#include <random>

#include "teaser/fpfh.h"
#include "teaser/matcher.h"
#include "teaser/ply_io.h"
#include "teaser/registration.h"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  teaser::PLYReader reader;
  teaser::PointCloud src_cloud;
  if (reader.read(argv[1], src_cloud) != 0 || src_cloud.size() < 10) return 2;
  const double T[3][4] = {{9.96926560e-01, 6.68735757e-02, -4.06664421e-02, -1.15576939e-01},
                          {-6.61289946e-02, 9.97617877e-01, 1.94008687e-02, -3.87705398e-02},
                          {4.18675510e-02, -1.66517807e-02, 9.98977765e-01, 1.14874890e-01}};
  std::mt19937 rng(7);
  std::uniform_real_distribution<double> noise(-0.0005, 0.0005);
  teaser::PointCloud tgt_cloud;
  for (size_t i = 0; i < src_cloud.size(); ++i) {
    const teaser::PointXYZ& p = src_cloud[i];
    double q[3];
    for (int r = 0; r < 3; ++r) q[r] = T[r][0] * p.x + T[r][1] * p.y + T[r][2] * p.z + T[r][3] + noise(rng);
    tgt_cloud.push_back({(float)q[0], (float)q[1], (float)q[2]});
  }
  try {
    teaser::FPFHEstimation fpfh;
    auto obj_descriptors = fpfh.computeFPFHFeatures(src_cloud, 0.02, 0.04);
    auto scene_descriptors = fpfh.computeFPFHFeatures(tgt_cloud, 0.02, 0.04);
    teaser::Matcher matcher;
    auto correspondences = matcher.calculateCorrespondences(src_cloud, tgt_cloud, *obj_descriptors, *scene_descriptors,
                                                            false, true, false, 0.95);
    teaser::RobustRegistrationSolver::Params params;
    params.noise_bound = 0.001;
    params.cbar2 = 1;
    params.estimate_scaling = false;
    params.rotation_max_iterations = 100;
    params.rotation_gnc_factor = 1.4;
    params.rotation_estimation_algorithm = teaser::RobustRegistrationSolver::ROTATION_ESTIMATION_ALGORITHM::GNC_TLS;
    params.rotation_cost_threshold = 0.005;
    teaser::RobustRegistrationSolver solver(params);
    solver.solve(src_cloud, tgt_cloud, correspondences);
    auto solution = solver.getSolution();
    double dR = 0, dt = 0;
    for (int r = 0; r < 3; ++r) {
      dt += std::pow(solution.translation(r) - T[r][3], 2);
      for (int c = 0; c < 3; ++c) dR += std::pow(solution.rotation(r, c) - T[r][c], 2);
    }
    std::printf("points %zu  correspondences %zu  clique %zu  valid %d  |R-R*|_F %.3e  |t-t*| %.3e  normal0 %.3f\n",
                src_cloud.size(), correspondences.size(), solver.getInlierMaxClique().size(), (int)solution.valid,
                std::sqrt(dR), std::sqrt(dt), fpfh.getNormals()[0].normal_z);
    return (solution.valid && std::sqrt(dR) < 0.02 && std::sqrt(dt) < 0.01 && correspondences.size() > 50) ? 0 : 1;
  } catch (const std::runtime_error& e) {
    std::printf("facade: %s\n", e.what());
    return 77;
  }
}
