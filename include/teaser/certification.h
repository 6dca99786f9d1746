// teaser/certification.h -- drop-in for the reference's teaser/include/teaser/certification.h:
// teaser::CertificationResult (:21-25), teaser::AbstractRotationCertifier (:27-50) and teaser::DRSCertifier
// (:53-239, teaser/src/certification.cc:22-190) over the MI355X C ABI (teaser_hip_certify in
// include/teaser_hip.h).  The stage functions the reference exposes for its unit tests (getQCost, getOmega1,
// getLambdaGuess, getLinearProjection, getOptimalDualProjection ...) are internal to the device
// implementation: the inverse map is never materialised and M_init only as its non-zero blocks
// (csrc/kernels_certify.hip, csrc/cert_setup.h).  No CPU path: certify() throws without an MI355X.
#pragma once

#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#if !defined(TEASER_HIP_NO_EIGEN) && (defined(TEASER_HIP_USE_EIGEN) || __has_include(<Eigen/Core>))
#include <Eigen/Core>
#define TEASER_HIP_CERT_HAVE_EIGEN 1
#else
#define TEASER_HIP_CERT_HAVE_EIGEN 0
#endif

#include "teaser_hip.h"

namespace teaser {

struct CertificationResult {  // certification.h:21-25
  bool is_optimal = false;
  double best_suboptimality = -1;
  std::vector<double> suboptimality_traj;
};

class DRSCertifier {
 public:
  enum class EIG_SOLVER_TYPE { EIGEN = 0, SPECTRA = 1 };  // accepted; the dense device solver is always used
  struct Params {  // certification.h:71-104: same fields, order and defaults
    double noise_bound = 0.01;
    double cbar2 = 1;
    double sub_optimality = 1e-3;
    double max_iterations = 2e2;
    double gamma_tau = 1.999999;
    EIG_SOLVER_TYPE eig_decomposition_solver = EIG_SOLVER_TYPE::EIGEN;
  };

  DRSCertifier() = delete;
  DRSCertifier(const Params& params) : params_(params) {}
  DRSCertifier(double noise_bound, double cbar2) {  // certification.h:116-119
    params_.noise_bound = noise_bound;
    params_.cbar2 = cbar2;
  }
  DRSCertifier(const DRSCertifier&) = delete;
  DRSCertifier& operator=(const DRSCertifier&) = delete;
  ~DRSCertifier() {
    if (h_) teaser_hip_solver_destroy(h_);
  }

  // R row-major 3 x 3; src / dst: n points, xyz interleaved (the memory of the reference's 3 x N column-major
  // matrices); theta: +1 inlier / -1 outlier.
  CertificationResult certify(const double* R, const double* src, const double* dst, const double* theta, int n) {
    if (!h_) {
      teaser_params_c c;
      teaser_hip_params_default(&c);
      const int32_t rc = teaser_hip_solver_create(&c, /*device=*/-1, &h_);
      if (rc != TEASER_HIP_OK) {
        h_ = nullptr;
        throw std::runtime_error("teaser::DRSCertifier: teaser_hip_solver_create failed (status " +
                                 std::to_string(rc) + "; 3 = no HIP device)");
      }
    }
    teaser_certifier_params_c p;
    p.noise_bound = params_.noise_bound;
    p.cbar2 = params_.cbar2;
    p.sub_optimality = params_.sub_optimality;
    p.max_iterations = params_.max_iterations;
    p.gamma_tau = params_.gamma_tau;
    teaser_certification_c out;
    std::vector<double> traj((size_t)(params_.max_iterations > 1 ? params_.max_iterations : 1));
    const int32_t rc = teaser_hip_certify(h_, &p, R, src, dst, theta, n, &out, traj.data(), (int32_t)traj.size());
    if (rc != TEASER_HIP_OK)
      throw std::runtime_error(std::string("teaser_hip_certify status ") + std::to_string(rc) + ": " +
                               teaser_hip_last_error(h_));
    CertificationResult r;
    r.is_optimal = out.is_optimal != 0;
    r.best_suboptimality = out.best_suboptimality;
    traj.resize((size_t)out.iterations);
    r.suboptimality_traj = traj;
    return r;
  }

#if TEASER_HIP_CERT_HAVE_EIGEN
  // the reference's signatures (certification.h:133-160)
  CertificationResult certify(const Eigen::Matrix3d& R_solution, const Eigen::Matrix<double, 3, Eigen::Dynamic>& src,
                              const Eigen::Matrix<double, 3, Eigen::Dynamic>& dst,
                              const Eigen::Matrix<double, 1, Eigen::Dynamic>& theta) {
    double R[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) R[3 * r + c] = R_solution(r, c);
    const Eigen::Matrix<double, 3, Eigen::Dynamic> a = src, b = dst;  // contiguous column-major copies
    const Eigen::Matrix<double, 1, Eigen::Dynamic> t = theta;
    return certify(R, a.data(), b.data(), t.data(), (int)a.cols());
  }
  CertificationResult certify(const Eigen::Matrix3d& R_solution, const Eigen::Matrix<double, 3, Eigen::Dynamic>& src,
                              const Eigen::Matrix<double, 3, Eigen::Dynamic>& dst,
                              const Eigen::Matrix<bool, 1, Eigen::Dynamic>& mask) {  // certification.cc:22-37
    Eigen::Matrix<double, 1, Eigen::Dynamic> theta(1, mask.cols());
    for (Eigen::Index i = 0; i < mask.cols(); ++i) theta(0, i) = mask(0, i) ? 1.0 : -1.0;
    return certify(R_solution, src, dst, theta);
  }
#endif

 private:
  Params params_;
  teaser_hip_solver* h_ = nullptr;
};

}  // namespace teaser
