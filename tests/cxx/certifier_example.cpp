// teaser/certification.h through the facade: a noiseless 8-point instance with the true rotation must be
// certified optimal (reference teaser/include/teaser/certification.h:53-239).  Exit code 77 without a GPU.
#include <cmath>
#include <cstdio>
#include <vector>

#include "teaser/certification.h"

int main() {
  const int n = 8;
  const double ang = 0.7, c = std::cos(ang), s = std::sin(ang);
  const double R[9] = {c, -s, 0, s, c, 0, 0, 0, 1};  // row-major
  std::vector<double> src, dst, theta;
  for (int k = 0; k < n; ++k) {
    const double p[3] = {std::sin(1.0 + k), std::cos(2.0 * k), std::sin(0.5 * k + 0.3)};
    for (double v : p) src.push_back(v);
    for (int r = 0; r < 3; ++r) dst.push_back(R[3 * r] * p[0] + R[3 * r + 1] * p[1] + R[3 * r + 2] * p[2]);
    theta.push_back(1.0);
  }
  // two gross outliers, flagged as such
  for (int k : {2, 5}) {
    dst[3 * k] += 0.9;
    dst[3 * k + 1] -= 0.7;
    theta[k] = -1.0;
  }
  teaser::DRSCertifier::Params params;
  params.noise_bound = 0.01;
  params.max_iterations = 50;
  try {
    teaser::DRSCertifier certifier(params);
    const teaser::CertificationResult res = certifier.certify(R, src.data(), dst.data(), theta.data(), n);
    std::printf("certifier: optimal %d, best sub-optimality %.3e after %zu iterations\n", (int)res.is_optimal,
                res.best_suboptimality, res.suboptimality_traj.size());
    return (res.is_optimal && !res.suboptimality_traj.empty()) ? 0 : 1;
  } catch (const std::runtime_error& e) {
    std::printf("certifier: %s\n", e.what());
    return 77;
  }
}
