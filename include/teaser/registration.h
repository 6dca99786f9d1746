// teaser/registration.h -- drop-in C++ facade: teaser::RobustRegistrationSolver over the MI355X C ABI
// (include/teaser_hip.h -> libteaser_hip.so).  Header-only; the numeric work is in the HIP library.
//
// Surface mirrored (names, field order, defaults, enum values) from the reference
// teaser/include/teaser/registration.h: RegistrationSolution :32-39, the stage-solver classes :40-360
// (AbstractScaleSolver / AbstractRotationSolver / AbstractTranslationSolver, ScalarTLSEstimator,
// TLSScaleSolver, ScaleInliersSelector, TLSTranslationSolver, GNCRotationSolver{Params},
// GNCTLSRotationSolver, FastGlobalRegistrationSolver, QuatroSolver), enums :382-412, Params :419-514,
// constructors :516-548, computeTIMs :555-557, solve :567-577, solveForScale / Rotation / Translation
// :584-601, set*Estimator :623-644, getters :609-824, reset :830-908, getParams :914.
//
// With Eigen available (<Eigen/Core> found, or TEASER_HIP_USE_EIGEN defined) the matrix types ARE the
// reference's Eigen types (Matrix<double,3,Dynamic>, Matrix<bool,1,Dynamic>, Matrix<int,2,Dynamic>, ...),
// so existing call sites compile unchanged and src.data() is handed to the device zero-copy.  Without
// Eigen (this repo's image has none) the same members are small value types with the accessors the
// reference's examples and tests use (operator()(r,c), operator()(i), data(), cols(), rows(), size()).
//
// Differences, all documented in DESIGN.md: the object is reusable (the reference object is
// single-use, registration.cc:702-704); `inlier_selection_mode` / `rotation_tim_graph` are honoured
// (the reference snapshot never stores params_); the M-sized products (getSrcTIMs, getScaleInliersMask,
// ...) are not kept on the device -- they are rebuilt on request, in the reference's pair order
// (registration.cc:531), from the inputs and the inlier graph; the constructor throws when no MI355X is
// visible (there is no CPU path), solve() itself never throws: a failed call returns valid = false and
// lastStatus() / lastError() say why.
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#if !defined(TEASER_HIP_NO_EIGEN) && (defined(TEASER_HIP_USE_EIGEN) || __has_include(<Eigen/Core>))
#include <Eigen/Core>
#define TEASER_HIP_HAVE_EIGEN 1
#else
#define TEASER_HIP_HAVE_EIGEN 0
#endif

#include "teaser/geometry.h"
#include "teaser/graph.h"
#include "teaser_hip.h"

namespace teaser {

#if TEASER_HIP_HAVE_EIGEN
using Matrix3X = Eigen::Matrix<double, 3, Eigen::Dynamic>;
using Matrix3 = Eigen::Matrix3d;
using Vector3 = Eigen::Vector3d;
using RowVectorXb = Eigen::Matrix<bool, 1, Eigen::Dynamic>;
using RowVectorXi = Eigen::Matrix<int, 1, Eigen::Dynamic>;
using RowVectorXd = Eigen::RowVectorXd;
using Matrix2Xi = Eigen::Matrix<int, 2, Eigen::Dynamic>;
#else
// column-major 3 x N doubles, same memory layout as Eigen::Matrix<double, 3, Eigen::Dynamic>
class Matrix3X {
 public:
  Matrix3X() = default;
  Matrix3X(int rows, int64_t cols) : d_((size_t)(3 * cols)) { (void)rows; }
  void resize(int rows, int64_t cols) { (void)rows; d_.assign((size_t)(3 * cols), 0.0); }
  int64_t cols() const { return (int64_t)(d_.size() / 3); }
  int rows() const { return 3; }
  double& operator()(int r, int64_t c) { return d_[(size_t)(3 * c + r)]; }
  double operator()(int r, int64_t c) const { return d_[(size_t)(3 * c + r)]; }
  double* data() { return d_.data(); }
  const double* data() const { return d_.data(); }

 private:
  std::vector<double> d_;
};
struct Matrix3 {  // column-major 3 x 3, like Eigen::Matrix3d
  std::array<double, 9> v{};
  double& operator()(int r, int c) { return v[(size_t)(3 * c + r)]; }
  double operator()(int r, int c) const { return v[(size_t)(3 * c + r)]; }
  const double* data() const { return v.data(); }
};
struct Vector3 {
  std::array<double, 3> v{};
  double& operator()(int r) { return v[(size_t)r]; }
  double operator()(int r) const { return v[(size_t)r]; }
  double& operator[](int r) { return v[(size_t)r]; }
  double operator[](int r) const { return v[(size_t)r]; }
  const double* data() const { return v.data(); }
};
// 1 x N row vector (Eigen::Matrix<T, 1, Dynamic> stand-in); bool is stored one byte per element
template <class T, class Store = T>
class RowVectorT {
 public:
  RowVectorT() = default;
  RowVectorT(int rows, int64_t cols) : d_((size_t)cols) { (void)rows; }
  explicit RowVectorT(int64_t cols) : d_((size_t)cols) {}
  void resize(int rows, int64_t cols) { (void)rows; d_.assign((size_t)cols, Store()); }
  int64_t cols() const { return (int64_t)d_.size(); }
  int64_t size() const { return (int64_t)d_.size(); }
  int rows() const { return 1; }
  Store& operator()(int64_t i) { return d_[(size_t)i]; }
  T operator()(int64_t i) const { return (T)d_[(size_t)i]; }
  Store& operator[](int64_t i) { return d_[(size_t)i]; }
  T operator[](int64_t i) const { return (T)d_[(size_t)i]; }
  Store* data() { return d_.data(); }
  const Store* data() const { return d_.data(); }
  operator std::vector<T>() const { return std::vector<T>(d_.begin(), d_.end()); }
  bool operator==(const RowVectorT& o) const { return d_ == o.d_; }

 private:
  std::vector<Store> d_;
};
using RowVectorXb = RowVectorT<bool, uint8_t>;
using RowVectorXi = RowVectorT<int>;
using RowVectorXd = RowVectorT<double>;
// column-major 2 x N ints (Eigen::Matrix<int, 2, Dynamic> stand-in)
class Matrix2Xi {
 public:
  Matrix2Xi() = default;
  Matrix2Xi(int rows, int64_t cols) : d_((size_t)(2 * cols)) { (void)rows; }
  void resize(int rows, int64_t cols) { (void)rows; d_.assign((size_t)(2 * cols), 0); }
  int64_t cols() const { return (int64_t)(d_.size() / 2); }
  int rows() const { return 2; }
  int& operator()(int r, int64_t c) { return d_[(size_t)(2 * c + r)]; }
  int operator()(int r, int64_t c) const { return d_[(size_t)(2 * c + r)]; }
  int* data() { return d_.data(); }
  const int* data() const { return d_.data(); }

 private:
  std::vector<int> d_;
};
#endif

struct RegistrationSolution {  // registration.h:32-39
  bool valid = true;
  double scale;
  Vector3 translation;
  Matrix3 rotation;
};

namespace detail {
// One process-wide "stage" handle per thread for the stand-alone stage-solver classes below (they carry
// only their parameters, like the reference's; the device context lives here).  Throws without a GPU.
inline teaser_hip_solver* stage_handle() {
  struct Holder {
    teaser_hip_solver* h = nullptr;
    ~Holder() {
      if (h) teaser_hip_solver_destroy(h);
    }
  };
  static thread_local Holder holder;
  if (!holder.h) {
    const int32_t rc = teaser_hip_solver_create(nullptr, /*device=*/-1, &holder.h);
    if (rc != TEASER_HIP_OK) {
      holder.h = nullptr;
      throw std::runtime_error("teaser stage solver: teaser_hip_solver_create failed (status " + std::to_string(rc) +
                               "; 3 = no HIP device)");
    }
  }
  return holder.h;
}
inline void stage_params(teaser_hip_solver* h, double noise_bound, double cbar2, bool estimate_scaling, int alg,
                         double gnc_factor, size_t max_iterations, double cost_threshold) {
  teaser_params_c c;
  teaser_hip_params_default(&c);
  c.noise_bound = noise_bound;
  c.cbar2 = cbar2;
  c.estimate_scaling = estimate_scaling ? 1 : 0;
  c.rotation_estimation_algorithm = alg;
  c.rotation_gnc_factor = gnc_factor;
  c.rotation_max_iterations = (int64_t)max_iterations;
  c.rotation_cost_threshold = cost_threshold;
  if (teaser_hip_solver_reset(h, &c) != TEASER_HIP_OK) throw std::runtime_error("teaser stage solver: reset failed");
}
inline void stage_check(teaser_hip_solver* h, int32_t rc) {
  if (rc != TEASER_HIP_OK)
    throw std::runtime_error(std::string("teaser_hip status ") + std::to_string(rc) + ": " + teaser_hip_last_error(h));
}
template <class Row>
inline void fill_mask(Row* out, const std::vector<uint8_t>& m) {
  if (!out) return;
  out->resize(1, (int64_t)m.size());
  for (size_t i = 0; i < m.size(); ++i) (*out)(i) = m[i] != 0;
}
}  // namespace detail

// ---- stage solvers (registration.h:40-360): same classes, the work runs on the GPU ---------------------
class AbstractScaleSolver {  // :40-53
 public:
  virtual ~AbstractScaleSolver() {}
  virtual void solveForScale(const Matrix3X& src, const Matrix3X& dst, double* scale, RowVectorXb* inliers) = 0;
};
class AbstractRotationSolver {  // :55-69
 public:
  virtual ~AbstractRotationSolver() {}
  virtual void solveForRotation(const Matrix3X& src, const Matrix3X& dst, Matrix3* rotation,
                                RowVectorXb* inliers) = 0;
};
class AbstractTranslationSolver {  // :71-85
 public:
  virtual ~AbstractTranslationSolver() {}
  virtual void solveForTranslation(const Matrix3X& src, const Matrix3X& dst, Vector3* translation,
                                   RowVectorXb* inliers) = 0;
};

class ScalarTLSEstimator {  // :87-115, registration.cc:21-88
 public:
  ScalarTLSEstimator() = default;
  void estimate(const RowVectorXd& X, const RowVectorXd& ranges, double* estimate, RowVectorXb* inliers) {
    teaser_hip_solver* h = detail::stage_handle();
    std::vector<uint8_t> m((size_t)X.cols());
    detail::stage_check(h, teaser_hip_scalar_tls(h, X.data(), ranges.data(), (int32_t)X.cols(), estimate, m.data()));
    detail::fill_mask(inliers, m);
  }
  // the tiled variant of the reference (registration.cc:90-204) computes the same estimate
  void estimate_tiled(const RowVectorXd& X, const RowVectorXd& ranges, const int& /*s*/, double* est,
                      RowVectorXb* inliers) {
    estimate(X, ranges, est, inliers);
  }
};

class TLSScaleSolver : public AbstractScaleSolver {  // :117-149, registration.cc:410-425
 public:
  TLSScaleSolver() = delete;
  explicit TLSScaleSolver(double noise_bound, double cbar2) : noise_bound_(noise_bound), cbar2_(cbar2) {}
  void solveForScale(const Matrix3X& src, const Matrix3X& dst, double* scale, RowVectorXb* inliers) override {
    teaser_hip_solver* h = detail::stage_handle();
    detail::stage_params(h, noise_bound_, cbar2_, true, 0, 1.4, 100, 1e-6);
    std::vector<uint8_t> m((size_t)src.cols());
    detail::stage_check(h, teaser_hip_solve_for_scale(h, src.data(), dst.data(), src.cols(), scale, m.data()));
    detail::fill_mask(inliers, m);
  }

 private:
  double noise_bound_, cbar2_;
};

class ScaleInliersSelector : public AbstractScaleSolver {  // :151-180, registration.cc:427-443
 public:
  ScaleInliersSelector() = delete;
  explicit ScaleInliersSelector(double noise_bound, double cbar2) : noise_bound_(noise_bound), cbar2_(cbar2) {}
  void solveForScale(const Matrix3X& src, const Matrix3X& dst, double* scale, RowVectorXb* inliers) override {
    teaser_hip_solver* h = detail::stage_handle();
    detail::stage_params(h, noise_bound_, cbar2_, false, 0, 1.4, 100, 1e-6);
    std::vector<uint8_t> m((size_t)src.cols());
    detail::stage_check(h, teaser_hip_solve_for_scale(h, src.data(), dst.data(), src.cols(), scale, m.data()));
    detail::fill_mask(inliers, m);
  }

 private:
  double noise_bound_, cbar2_;
};

class TLSTranslationSolver : public AbstractTranslationSolver {  // :182-213, registration.cc:445-471
 public:
  TLSTranslationSolver() = delete;
  explicit TLSTranslationSolver(double noise_bound, double cbar2) : noise_bound_(noise_bound), cbar2_(cbar2) {}
  void solveForTranslation(const Matrix3X& src, const Matrix3X& dst, Vector3* translation,
                           RowVectorXb* inliers) override {
    teaser_hip_solver* h = detail::stage_handle();
    detail::stage_params(h, noise_bound_, cbar2_, false, 0, 1.4, 100, 1e-6);
    std::vector<uint8_t> m((size_t)src.cols());
    double t[3];
    detail::stage_check(h, teaser_hip_solve_for_translation(h, src.data(), dst.data(), (int32_t)src.cols(), t, m.data()));
    if (translation)
      for (int r = 0; r < 3; ++r) (*translation)(r) = t[r];
    detail::fill_mask(inliers, m);
  }

 private:
  double noise_bound_, cbar2_;
};

class GNCRotationSolver : public AbstractRotationSolver {  // :215-244
 public:
  struct Params {  // :223-228
    size_t max_iterations;
    double cost_threshold;
    double gnc_factor;
    double noise_bound;
  };
  GNCRotationSolver(Params params) : params_(params) {}
  Params getParams() { return params_; }
  void setParams(Params params) { params_ = params; }
  double getCostAtTermination() { return cost_; }

 protected:
  // algorithm: TEASER_ROT_GNC_TLS / _FGR / _QUATRO
  void run(int algorithm, const Matrix3X& src, const Matrix3X& dst, Matrix3* rotation, RowVectorXb* inliers) {
    teaser_hip_solver* h = detail::stage_handle();
    detail::stage_params(h, params_.noise_bound, 1.0, false, algorithm, params_.gnc_factor, params_.max_iterations,
                         params_.cost_threshold);
    std::vector<uint8_t> m((size_t)src.cols());
    double R[9], cost = 0;
    int32_t iters = 0;
    detail::stage_check(h, teaser_hip_solve_for_rotation(h, src.data(), dst.data(), (int32_t)src.cols(),
                                                         params_.noise_bound, R, m.data(), &cost, &iters));
    cost_ = cost;
    if (rotation)
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) (*rotation)(r, c) = R[3 * r + c];
    detail::fill_mask(inliers, m);
  }
  Params params_;
  double cost_ = 0;
};
class GNCTLSRotationSolver : public GNCRotationSolver {  // :246-280, registration.cc:764-866
 public:
  GNCTLSRotationSolver() = delete;
  explicit GNCTLSRotationSolver(Params params) : GNCRotationSolver(params) {}
  void solveForRotation(const Matrix3X& src, const Matrix3X& dst, Matrix3* rotation, RowVectorXb* inliers) override {
    run(TEASER_ROT_GNC_TLS, src, dst, rotation, inliers);
  }
};
class FastGlobalRegistrationSolver : public GNCRotationSolver {  // :282-320, registration.cc:206-278
 public:
  FastGlobalRegistrationSolver() = delete;
  explicit FastGlobalRegistrationSolver(Params params) : GNCRotationSolver(params) {}
  void solveForRotation(const Matrix3X& src, const Matrix3X& dst, Matrix3* rotation, RowVectorXb* inliers) override {
    run(TEASER_ROT_FGR, src, dst, rotation, inliers);
  }
};
class QuatroSolver : public GNCRotationSolver {  // :322-359, registration.cc:280-408
 public:
  QuatroSolver() = delete;
  explicit QuatroSolver(Params params) : GNCRotationSolver(params) {}
  void solveForRotation(const Matrix3X& src, const Matrix3X& dst, Matrix3* rotation, RowVectorXb* inliers) override {
    run(TEASER_ROT_QUATRO, src, dst, rotation, inliers);
  }
};

class RobustRegistrationSolver {
 public:
  enum class ROTATION_ESTIMATION_ALGORITHM { GNC_TLS = 0, FGR = 1, QUATRO = 2 };           // :382-386
  enum class INLIER_SELECTION_MODE { PMC_EXACT = 0, PMC_HEU = 1, KCORE_HEU = 2, NONE = 3 };  // :396-401
  enum class INLIER_GRAPH_FORMULATION { CHAIN = 0, COMPLETE = 1 };                           // :409-412

  struct Params {  // registration.h:419-514: same fields, order and defaults
    double noise_bound = 0.01;
    double cbar2 = 1;
    bool estimate_scaling = true;
    ROTATION_ESTIMATION_ALGORITHM rotation_estimation_algorithm = ROTATION_ESTIMATION_ALGORITHM::GNC_TLS;
    double rotation_gnc_factor = 1.4;
    size_t rotation_max_iterations = 100;
    double rotation_cost_threshold = 1e-6;
    INLIER_GRAPH_FORMULATION rotation_tim_graph = INLIER_GRAPH_FORMULATION::CHAIN;
    INLIER_SELECTION_MODE inlier_selection_mode = INLIER_SELECTION_MODE::PMC_EXACT;
    double kcore_heuristic_threshold = 0.5;
    bool use_max_clique = true;             // deprecated in the reference, still honoured (:574-578)
    bool max_clique_exact_solution = true;  // deprecated in the reference, still honoured (:579-583)
    double max_clique_time_limit = 3600;
    int max_clique_num_threads = 0;  // accepted; the GPU search is not thread-count parameterised
  };

  RobustRegistrationSolver() { reset(Params()); }                       // :516
  explicit RobustRegistrationSolver(const Params& params) { reset(params); }  // :548
  RobustRegistrationSolver(double noise_bound, double cbar2, bool estimate_scaling,  // :524-539
                           ROTATION_ESTIMATION_ALGORITHM rotation_estimation_algorithm,
                           double rotation_gnc_factor, size_t rotation_max_iterations,
                           double rotation_cost_threshold, INLIER_GRAPH_FORMULATION rotation_tim_graph,
                           INLIER_SELECTION_MODE inlier_selection_mode,
                           double kcore_heuristic_threshold, bool use_max_clique,
                           bool max_clique_exact_solution, double max_clique_time_limit,
                           int max_clique_num_threads = 0) {
    Params p;
    p.noise_bound = noise_bound;
    p.cbar2 = cbar2;
    p.estimate_scaling = estimate_scaling;
    p.rotation_estimation_algorithm = rotation_estimation_algorithm;
    p.rotation_gnc_factor = rotation_gnc_factor;
    p.rotation_max_iterations = rotation_max_iterations;
    p.rotation_cost_threshold = rotation_cost_threshold;
    p.rotation_tim_graph = rotation_tim_graph;
    p.inlier_selection_mode = inlier_selection_mode;
    p.kcore_heuristic_threshold = kcore_heuristic_threshold;
    p.use_max_clique = use_max_clique;
    p.max_clique_exact_solution = max_clique_exact_solution;
    p.max_clique_time_limit = max_clique_time_limit;
    p.max_clique_num_threads = max_clique_num_threads;
    reset(p);
  }
  RobustRegistrationSolver(const RobustRegistrationSolver&) = delete;
  RobustRegistrationSolver& operator=(const RobustRegistrationSolver&) = delete;
  ~RobustRegistrationSolver() {
    if (h_) teaser_hip_solver_destroy(h_);
  }

  // registration.h:891.  Throws std::runtime_error when no MI355X is visible: the product has no
  // CPU path (the reference's constructor cannot fail; a host without a GPU must say so loudly).
  // Like the reference's reset (:830-885) this re-creates the default stage solvers from the Params:
  // custom estimators installed with set*Estimator are dropped.
  void reset(const Params& params) {
    params_ = params;
    const teaser_params_c c = to_c(params);
    if (!h_) {
      const int32_t rc = teaser_hip_solver_create(&c, /*device=*/-1, &h_);
      if (rc != TEASER_HIP_OK) {
        h_ = nullptr;
        throw std::runtime_error("teaser::RobustRegistrationSolver: teaser_hip_solver_create failed (status " +
                                 std::to_string(rc) + "; 3 = no HIP device)");
      }
    } else {
      check(teaser_hip_solver_reset(h_, &c));
    }
    scale_solver_.reset();
    rotation_solver_.reset();
    translation_solver_.reset();
    solution_ = RegistrationSolution();
    have_solution_ = false;
    max_clique_.clear();
    rotation_inliers_.clear();
    translation_inliers_.clear();
    last_status_ = TEASER_HIP_OK;
    last_error_.clear();
  }
  Params getParams() { return params_; }  // :914

  // registration.h:555-557, registration.cc:512-551: all pairwise differences v_j - v_i, j > i, in the
  // pair order k = i N - i (i + 1) / 2 + (j - i - 1); map(:, k) = (i, j).  Host-side helper (the solve
  // path never materialises TIMs: this exists for callers of the reference's public method).
  Matrix3X computeTIMs(const Matrix3X& v, Matrix2Xi* map) {
    const int64_t N = v.cols(), M = N * (N - 1) / 2;
    Matrix3X tims(3, M > 0 ? M : 0);
    if (map) map->resize(2, M > 0 ? M : 0);
    int64_t k = 0;
    for (int64_t i = 0; i + 1 < N; ++i)
      for (int64_t j = i + 1; j < N; ++j, ++k) {
        for (int r = 0; r < 3; ++r) tims(r, k) = v(r, j) - v(r, i);
        if (map) {
          (*map)(0, k) = (int)i;
          (*map)(1, k) = (int)j;
        }
      }
    return tims;
  }

  // registration.h:576-577 -- 3 x N matrices of corresponding points (column-major doubles).  Never
  // throws: on failure the returned solution has valid = false and lastStatus() / lastError() are set.
  RegistrationSolution solve(const Matrix3X& src, const Matrix3X& dst) {
    last_status_ = TEASER_HIP_OK;
    last_error_.clear();
    if (src.cols() != dst.cols()) return fail(TEASER_HIP_ERR_BAD_ARG, "solve: src and dst differ in size");
    keep_inputs(src, dst);
    if (scale_solver_ || rotation_solver_ || translation_solver_) return solve_staged();
    teaser_solution_c o;
    const int32_t rc = teaser_hip_solve(h_, src.data(), dst.data(), (int32_t)src.cols(), &o);
    if (rc != TEASER_HIP_OK && rc != TEASER_HIP_ERR_TIME_LIMIT) return fail(rc, teaser_hip_last_error(h_));
    last_status_ = rc;
    adopt(o);
    fetch_lists();
    return solution_;
  }
  // registration.h:567-569, registration.cc:553-566 -- clouds + correspondences (src index, dst index)
  RegistrationSolution solve(const PointCloud& src_cloud, const PointCloud& dst_cloud,
                             const std::vector<std::pair<int, int>> correspondences) {
    const int64_t C = (int64_t)correspondences.size();
    Matrix3X s(3, C), d(3, C);
    for (int64_t i = 0; i < C; ++i) {
      const int a = correspondences[(size_t)i].first, b = correspondences[(size_t)i].second;
      if (a < 0 || b < 0 || (size_t)a >= src_cloud.size() || (size_t)b >= dst_cloud.size())
        return fail(TEASER_HIP_ERR_BAD_ARG, "solve: correspondence index out of range");
      const PointXYZ& p = src_cloud[(size_t)a];
      const PointXYZ& q = dst_cloud[(size_t)b];
      s(0, i) = p.x; s(1, i) = p.y; s(2, i) = p.z;  // float -> double widening, :559-564
      d(0, i) = q.x; d(1, i) = q.y; d(2, i) = q.z;
    }
    return solve(s, d);
  }

  // Stage entry points (registration.h:584-601): 3 x K TIMs / points -> the stage's estimate with the
  // installed (default: GPU) estimator; the result is also stored in the solution, like the reference.
  double solveForScale(const Matrix3X& v1, const Matrix3X& v2) {  // :584, registration.cc:739-745
    if (v1.cols() != v2.cols()) throw std::invalid_argument("solveForScale: sizes differ");
    double scale = 1;
    if (scale_solver_) {
      scale_solver_->solveForScale(v1, v2, &scale, &scale_inliers_mask_);
    } else {
      std::vector<uint8_t> m((size_t)v1.cols());
      check(teaser_hip_solve_for_scale(h_, v1.data(), v2.data(), v1.cols(), &scale, m.data()));
      detail::fill_mask(&scale_inliers_mask_, m);
    }
    have_stage_scale_mask_ = true;
    solution_.scale = scale;
    return scale;
  }
  Matrix3 solveForRotation(const Matrix3X& v1, const Matrix3X& v2) {  // :593, registration.cc:756-762
    if (v1.cols() != v2.cols()) throw std::invalid_argument("solveForRotation: sizes differ");
    if (rotation_solver_) {
      rotation_solver_->solveForRotation(v1, v2, &solution_.rotation, &rotation_inliers_mask_);
      raw_.gnc_cost = rotation_solver_->getCostAtTermination();
    } else {
      double R[9], cost = 0;
      int32_t iters = 0;
      std::vector<uint8_t> m((size_t)v1.cols());
      check(teaser_hip_solve_for_rotation(h_, v1.data(), v2.data(), (int32_t)v1.cols(), params_.noise_bound, R,
                                          m.data(), &cost, &iters));
      raw_.gnc_cost = cost;
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) solution_.rotation(r, c) = R[3 * r + c];
      detail::fill_mask(&rotation_inliers_mask_, m);
    }
    have_stage_rotation_mask_ = true;
    return solution_.rotation;
  }
  Vector3 solveForTranslation(const Matrix3X& v1, const Matrix3X& v2) {  // :601, registration.cc:747-754
    if (v1.cols() != v2.cols()) throw std::invalid_argument("solveForTranslation: sizes differ");
    if (translation_solver_) {
      translation_solver_->solveForTranslation(v1, v2, &solution_.translation, &translation_inliers_mask_);
    } else {
      double t[3];
      std::vector<uint8_t> m((size_t)v1.cols());
      check(teaser_hip_solve_for_translation(h_, v1.data(), v2.data(), (int32_t)v1.cols(), t, m.data()));
      for (int r = 0; r < 3; ++r) solution_.translation(r) = t[r];
      detail::fill_mask(&translation_inliers_mask_, m);
    }
    have_stage_translation_mask_ = true;
    return solution_.translation;
  }
  // inlier mask of the last stage call above (one entry per column) -- kept from round 1
  std::vector<bool> getStageInliersMask() const {
    const RowVectorXb& m = have_stage_translation_mask_ ? translation_inliers_mask_ : rotation_inliers_mask_;
    std::vector<bool> out((size_t)m.cols());
    for (int64_t i = 0; i < m.cols(); ++i) out[(size_t)i] = m(i);
    return out;
  }

  // registration.h:623-644: custom estimators.  solveFor{Scale,Rotation,Translation} use them directly;
  // solve() then runs stage by stage (TIMs are materialised on the host only when a custom SCALE
  // estimator is installed -- O(N^2) memory, as in the reference).
  void setScaleEstimator(std::unique_ptr<AbstractScaleSolver> estimator) { scale_solver_ = std::move(estimator); }
  void setRotationEstimator(std::unique_ptr<GNCRotationSolver> estimator) { rotation_solver_ = std::move(estimator); }
  void setTranslationEstimator(std::unique_ptr<AbstractTranslationSolver> estimator) {
    translation_solver_ = std::move(estimator);
  }

  RegistrationSolution getSolution() { return solution_; }            // :617
  double getGNCRotationCostAtTermination() { return raw_.gnc_cost; }  // :609-611
  std::vector<int> getInlierMaxClique() { return max_clique_; }       // :770
  std::vector<int> getRotationInliers() { return rotation_inliers_; }  // :713
  std::vector<int> getTranslationInliers() { return translation_inliers_; }  // :744
  std::vector<int> getInputOrderedTranslationInliers() {                      // :752-763
    if (params_.rotation_estimation_algorithm == ROTATION_ESTIMATION_ALGORITHM::FGR)
      throw std::runtime_error(
          "This function is not supported when using FGR since FGR does not use max clique.");
    std::vector<int> out;
    out.reserve(translation_inliers_.size());
    for (int i : translation_inliers_) out.push_back(max_clique_[(size_t)i]);
    return out;
  }
  // masks over the rotation TIMs / the clique (registration.h:689, :723-725) and their maps (:698-737)
  RowVectorXb getRotationInliersMask() {
    if (have_stage_rotation_mask_) return rotation_inliers_mask_;
    return index_mask(rotation_inliers_, n_rotation_tims());
  }
  RowVectorXb getTranslationInliersMask() {
    if (have_stage_translation_mask_) return translation_inliers_mask_;
    return index_mask(translation_inliers_, max_clique_.size());
  }
  RowVectorXi getRotationInliersMap() { return clique_row(); }
  RowVectorXi getTranslationInliersMap() { return clique_row(); }
  // registration.h:772: adjacency list of the inlier graph, unpacked from the device's bit matrix
  std::vector<std::vector<int>> getInlierGraph() {
    if (staged_graph_valid_) return staged_graph_.getAdjList();
    const int n = raw_.n, W = (n + 63) / 64;
    std::vector<std::vector<int>> adj((size_t)(n > 0 ? n : 0));
    if (!have_solution_ || n <= 0) return adj;
    const std::vector<uint64_t> bm = graph_bitmap();
    if (bm.empty()) return adj;
    for (int i = 0; i < n; ++i)
      for (int w = 0; w < W; ++w) {
        uint64_t bits = bm[(size_t)i * (size_t)W + (size_t)w];
        while (bits) {
          adj[(size_t)i].push_back(64 * w + __builtin_ctzll(bits));
          bits &= bits - 1;
        }
      }
    return adj;
  }
  // registration.h:652: 1 x M mask of the TIMs that passed the scale stage, in the pair order of
  // computeTIMs; rebuilt from the inlier graph (edge (i, j) <=> TIM k(i, j) is a scale inlier, :614-619)
  RowVectorXb getScaleInliersMask() {
    if (have_stage_scale_mask_) return scale_inliers_mask_;
    const int64_t N = raw_.n, M = N * (N - 1) / 2, W = (N + 63) / 64;
    RowVectorXb mask(1, M > 0 ? M : 0);
    const std::vector<uint64_t> bm = have_solution_ ? graph_bitmap() : std::vector<uint64_t>();
    int64_t k = 0;
    for (int64_t i = 0; i + 1 < N; ++i)
      for (int64_t j = i + 1; j < N; ++j, ++k)
        mask(k) = !bm.empty() && ((bm[(size_t)(i * W + (j >> 6))] >> (j & 63)) & 1ull) != 0;
    return mask;
  }
  Matrix2Xi getScaleInliersMap() { return pair_map(raw_.n); }  // :662 (= the src TIM map)
  // registration.h:671-679: the TIMs (pairs of correspondences) that passed the scale stage
  std::vector<std::tuple<int, int>> getScaleInliers() {
    std::vector<std::tuple<int, int>> out;
    const auto adj = getInlierGraph();
    for (int i = 0; i < (int)adj.size(); ++i)
      for (int j : adj[(size_t)i])
        if (j > i) out.emplace_back(i, j);
    return out;
  }
  // registration.h:778-824: TIM products, rebuilt on request from the inputs of the last solve
  Matrix3X getSrcTIMs() { return computeTIMs(last_src_, nullptr); }
  Matrix3X getDstTIMs() { return computeTIMs(last_dst_, nullptr); }
  Matrix2Xi getSrcTIMsMap() { return pair_map(last_src_.cols()); }
  Matrix2Xi getDstTIMsMap() { return pair_map(last_dst_.cols()); }
  Matrix3X getMaxCliqueSrcTIMs() { return clique_tims(last_src_, 1.0, nullptr); }  // :790
  Matrix3X getMaxCliqueDstTIMs() { return clique_tims(last_dst_, 1.0 / solution_.scale, nullptr); }  // :796, :697
  Matrix2Xi getSrcTIMsMapForRotation() {  // :814
    Matrix2Xi m;
    clique_tims(last_src_, 1.0, &m);
    return m;
  }
  Matrix2Xi getDstTIMsMapForRotation() { return getSrcTIMsMapForRotation(); }  // :824

  // not part of the reference surface
  teaser_hip_solver* handle() { return h_; }
  const teaser_solution_c& rawSolution() const { return raw_; }
  int32_t lastStatus() const { return last_status_; }          // teaser_hip_status of the last solve()
  const std::string& lastError() const { return last_error_; }  // its message ("" when OK)

 private:
  static teaser_params_c to_c(const Params& p) {
    teaser_params_c c;
    teaser_hip_params_default(&c);
    c.noise_bound = p.noise_bound;
    c.cbar2 = p.cbar2;
    c.estimate_scaling = p.estimate_scaling ? 1 : 0;
    c.rotation_estimation_algorithm = (int32_t)p.rotation_estimation_algorithm;
    c.rotation_gnc_factor = p.rotation_gnc_factor;
    c.rotation_max_iterations = (int64_t)p.rotation_max_iterations;
    c.rotation_cost_threshold = p.rotation_cost_threshold;
    c.rotation_tim_graph = (int32_t)p.rotation_tim_graph;
    c.inlier_selection_mode = (int32_t)p.inlier_selection_mode;
    c.kcore_heuristic_threshold = p.kcore_heuristic_threshold;
    c.use_max_clique = p.use_max_clique ? 1 : 0;
    c.max_clique_exact_solution = p.max_clique_exact_solution ? 1 : 0;
    c.max_clique_time_limit = p.max_clique_time_limit;
    c.max_clique_num_threads = p.max_clique_num_threads;
    return c;
  }
  void check(int32_t rc) const {
    // TIME_LIMIT still carries the incumbent clique (graph.cc:44): not an error for the caller
    if (rc != TEASER_HIP_OK && rc != TEASER_HIP_ERR_TIME_LIMIT)
      throw std::runtime_error(std::string("teaser_hip status ") + std::to_string(rc) + ": " +
                               (h_ ? teaser_hip_last_error(h_) : ""));
  }
  RegistrationSolution fail(int32_t rc, const std::string& msg) {
    last_status_ = rc;
    last_error_ = msg;
    solution_ = RegistrationSolution();
    solution_.valid = false;
    solution_.scale = 1;
    have_solution_ = false;
    max_clique_.clear();
    rotation_inliers_.clear();
    translation_inliers_.clear();
    return solution_;
  }
  void keep_inputs(const Matrix3X& src, const Matrix3X& dst) {
    last_src_ = src;
    last_dst_ = dst;
    staged_graph_valid_ = false;
    have_stage_scale_mask_ = have_stage_rotation_mask_ = have_stage_translation_mask_ = false;
  }
  void adopt(const teaser_solution_c& o) {
    raw_ = o;
    have_solution_ = true;
    solution_.valid = o.valid != 0;
    solution_.scale = o.scale;
    for (int r = 0; r < 3; ++r) {
      solution_.translation(r) = o.translation[r];
      for (int c = 0; c < 3; ++c) solution_.rotation(r, c) = o.rotation[3 * r + c];  // ABI: row-major
    }
  }
  template <class F>
  std::vector<int> list(F getter) {
    std::vector<int> v;
    int64_t len = 0;
    if (getter(h_, 0, nullptr, &len) != TEASER_HIP_OK || len <= 0) return v;
    v.resize((size_t)len);
    if (getter(h_, 0, reinterpret_cast<int32_t*>(v.data()), &len) != TEASER_HIP_OK) v.clear();
    return v;
  }
  void fetch_lists() {
    max_clique_ = list(teaser_hip_get_max_clique);
    rotation_inliers_ = list(teaser_hip_get_rotation_inliers);
    translation_inliers_ = list(teaser_hip_get_translation_inliers);
  }
  std::vector<uint64_t> graph_bitmap() {
    int64_t len = 0;
    std::vector<uint64_t> bm;
    if (teaser_hip_get_inlier_graph_bitmap(h_, 0, nullptr, &len) != TEASER_HIP_OK || len <= 0) return bm;
    bm.resize((size_t)len);
    if (teaser_hip_get_inlier_graph_bitmap(h_, 0, bm.data(), &len) != TEASER_HIP_OK) bm.clear();
    return bm;
  }
  static RowVectorXb index_mask(const std::vector<int>& idx, size_t n) {
    RowVectorXb m(1, (int64_t)n);
    for (size_t i = 0; i < n; ++i) m((int64_t)i) = false;
    for (int i : idx)
      if (i >= 0 && (size_t)i < n) m(i) = true;
    return m;
  }
  RowVectorXi clique_row() const {
    RowVectorXi m(1, (int64_t)max_clique_.size());
    for (size_t i = 0; i < max_clique_.size(); ++i) m((int64_t)i) = max_clique_[i];
    return m;
  }
  static Matrix2Xi pair_map(int64_t N) {
    const int64_t M = N > 1 ? N * (N - 1) / 2 : 0;
    Matrix2Xi m(2, M);
    int64_t k = 0;
    for (int64_t i = 0; i + 1 < N; ++i)
      for (int64_t j = i + 1; j < N; ++j, ++k) {
        m(0, k) = (int)i;
        m(1, k) = (int)j;
      }
    return m;
  }
  size_t n_rotation_tims() const {  // CHAIN: K TIMs; COMPLETE: K (K - 1) / 2 (registration.cc:657-694)
    const size_t K = max_clique_.size();
    return params_.rotation_tim_graph == INLIER_GRAPH_FORMULATION::CHAIN ? K : K * (K - 1) / 2;
  }
  // TIMs on the clique for the rotation stage (registration.cc:657-694): CHAIN (i -> i + 1, wrapping)
  // or COMPLETE; `mul` = 1 / scale for the dst side (:697); map columns are (leaf, root) for CHAIN as the
  // reference writes them (:671-676), (i, j) clique positions mapped to input indices for COMPLETE.
  Matrix3X clique_tims(const Matrix3X& pts, double mul, Matrix2Xi* map) const {
    const int64_t K = (int64_t)max_clique_.size();
    if (K < 2 || pts.cols() == 0) {
      if (map) map->resize(2, 0);
      return Matrix3X(3, 0);
    }
    if (params_.rotation_tim_graph == INLIER_GRAPH_FORMULATION::CHAIN) {
      Matrix3X t(3, K);
      if (map) map->resize(2, K);
      for (int64_t i = 0; i < K; ++i) {
        const int root = max_clique_[(size_t)i], leaf = max_clique_[(size_t)((i + 1) % K)];
        for (int r = 0; r < 3; ++r) t(r, i) = (pts(r, leaf) - pts(r, root)) * mul;
        if (map) {
          (*map)(0, i) = leaf;
          (*map)(1, i) = root;
        }
      }
      return t;
    }
    const int64_t M = K * (K - 1) / 2;
    Matrix3X t(3, M);
    if (map) map->resize(2, M);
    int64_t k = 0;
    for (int64_t i = 0; i + 1 < K; ++i)
      for (int64_t j = i + 1; j < K; ++j, ++k) {
        const int a = max_clique_[(size_t)i], b = max_clique_[(size_t)j];
        for (int r = 0; r < 3; ++r) t(r, k) = (pts(r, b) - pts(r, a)) * mul;
        if (map) {
          (*map)(0, k) = a;
          (*map)(1, k) = b;
        }
      }
    return t;
  }

  // solve() with custom estimators installed: the stages of registration.cc:568-737 one by one.
  RegistrationSolution solve_staged() {
    try {
      const int64_t N = last_src_.cols();
      raw_ = teaser_solution_c();
      raw_.n = (int32_t)N;
      solution_ = RegistrationSolution();
      solution_.scale = 1;
      const bool use_clique = params_.use_max_clique && params_.inlier_selection_mode != INLIER_SELECTION_MODE::NONE;
      if (scale_solver_) {
        // :599-619 with the caller's scale estimator: TIMs and the graph on the host
        Matrix2Xi map;
        const Matrix3X st = computeTIMs(last_src_, &map), dt = computeTIMs(last_dst_, nullptr);
        double scale = 1;
        scale_solver_->solveForScale(st, dt, &scale, &scale_inliers_mask_);
        have_stage_scale_mask_ = true;
        solution_.scale = scale;
        staged_graph_.clear();
        staged_graph_.populateVertices((int)N);
        for (int64_t k = 0; k < scale_inliers_mask_.cols(); ++k)
          if (scale_inliers_mask_(k)) staged_graph_.addEdge(map(0, k), map(1, k));
        staged_graph_valid_ = true;
        if (use_clique) {
          MaxCliqueSolver::Params cp;  // :621-632
          cp.solver_mode = (MaxCliqueSolver::CLIQUE_SOLVER_MODE)(int)params_.inlier_selection_mode;
          cp.solve_exactly = params_.max_clique_exact_solution;
          cp.kcore_heuristic_threshold = params_.kcore_heuristic_threshold;
          cp.time_limit = params_.max_clique_time_limit;
          MaxCliqueSolver cs(cp);
          max_clique_ = cs.findMaxClique(staged_graph_);
        }
      } else {
        // default scale stage + graph + clique on the device; its rotation / translation are replaced below
        teaser_solution_c o;
        const int32_t rc = teaser_hip_solve(h_, last_src_.data(), last_dst_.data(), (int32_t)N, &o);
        if (rc != TEASER_HIP_OK && rc != TEASER_HIP_ERR_TIME_LIMIT) return fail(rc, teaser_hip_last_error(h_));
        adopt(o);
        fetch_lists();
      }
      if (!use_clique) {  // :648-654
        max_clique_.resize((size_t)N);
        for (int64_t i = 0; i < N; ++i) max_clique_[(size_t)i] = (int)i;
      }
      have_solution_ = true;
      raw_.clique_size = (int32_t)max_clique_.size();
      if (max_clique_.size() <= 1) {  // :643-647
        solution_.valid = false;
        rotation_inliers_.clear();
        translation_inliers_.clear();
        return solution_;
      }
      // :657-704 TIMs on the clique, de-scaled; rotation noise bound * 2 / scale
      const Matrix3X ps = clique_tims(last_src_, 1.0, nullptr), pd = clique_tims(last_dst_, 1.0 / solution_.scale, nullptr);
      if (rotation_solver_) {
        const GNCRotationSolver::Params keep = rotation_solver_->getParams();
        GNCRotationSolver::Params rp = keep;
        rp.noise_bound *= 2 / solution_.scale;
        rotation_solver_->setParams(rp);
        rotation_solver_->solveForRotation(ps, pd, &solution_.rotation, &rotation_inliers_mask_);
        rotation_solver_->setParams(keep);  // (the reference leaves the mutation in place: single-use object)
        raw_.gnc_cost = rotation_solver_->getCostAtTermination();
      } else {
        double R[9], cost = 0;
        int32_t iters = 0;
        std::vector<uint8_t> m((size_t)ps.cols());
        check(teaser_hip_solve_for_rotation(h_, ps.data(), pd.data(), (int32_t)ps.cols(),
                                            params_.noise_bound * 2 / solution_.scale, R, m.data(), &cost, &iters));
        raw_.gnc_cost = cost;
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) solution_.rotation(r, c) = R[3 * r + c];
        detail::fill_mask(&rotation_inliers_mask_, m);
      }
      have_stage_rotation_mask_ = true;
      rotation_inliers_.clear();  // :712-716
      for (int64_t i = 0; i < rotation_inliers_mask_.cols(); ++i)
        if (rotation_inliers_mask_(i)) rotation_inliers_.push_back((int)i);
      // :717-731 translation on the clique points: dst - s R src
      const int64_t K = (int64_t)max_clique_.size();
      Matrix3X a(3, K), b(3, K);
      for (int64_t i = 0; i < K; ++i) {
        const int v = max_clique_[(size_t)i];
        for (int r = 0; r < 3; ++r) {
          double acc = 0;
          for (int c = 0; c < 3; ++c) acc += solution_.scale * solution_.rotation(r, c) * last_src_(c, v);
          a(r, i) = acc;
          b(r, i) = last_dst_(r, v);
        }
      }
      if (translation_solver_) {
        translation_solver_->solveForTranslation(a, b, &solution_.translation, &translation_inliers_mask_);
      } else {
        double t[3];
        std::vector<uint8_t> m((size_t)K);
        check(teaser_hip_solve_for_translation(h_, a.data(), b.data(), (int32_t)K, t, m.data()));
        for (int r = 0; r < 3; ++r) solution_.translation(r) = t[r];
        detail::fill_mask(&translation_inliers_mask_, m);
      }
      have_stage_translation_mask_ = true;
      translation_inliers_.clear();
      for (int64_t i = 0; i < translation_inliers_mask_.cols(); ++i)
        if (translation_inliers_mask_(i)) translation_inliers_.push_back((int)i);
      solution_.valid = true;  // :734
      return solution_;
    } catch (const std::exception& e) {
      return fail(TEASER_HIP_ERR_HIP, e.what());
    }
  }

  teaser_hip_solver* h_ = nullptr;
  Params params_;
  RegistrationSolution solution_;
  teaser_solution_c raw_{};
  bool have_solution_ = false;
  int32_t last_status_ = TEASER_HIP_OK;
  std::string last_error_;
  std::vector<int> max_clique_, rotation_inliers_, translation_inliers_;
  Matrix3X last_src_, last_dst_;  // inputs of the last solve: the lazily rebuilt TIM products need them
  RowVectorXb scale_inliers_mask_, rotation_inliers_mask_, translation_inliers_mask_;
  bool have_stage_scale_mask_ = false, have_stage_rotation_mask_ = false, have_stage_translation_mask_ = false;
  Graph staged_graph_;
  bool staged_graph_valid_ = false;
  std::unique_ptr<AbstractScaleSolver> scale_solver_;
  std::unique_ptr<GNCRotationSolver> rotation_solver_;
  std::unique_ptr<AbstractTranslationSolver> translation_solver_;
};

}  // namespace teaser
