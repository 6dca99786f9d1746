// teaser/registration.h -- drop-in C++ facade: teaser::RobustRegistrationSolver over the MI355X C ABI
// (include/teaser_hip.h -> libteaser_hip.so).  Header-only; the numeric work is in the HIP library.
//
// Surface mirrored (names, field order, defaults, enum values) from the reference
// teaser/include/teaser/registration.h: RegistrationSolution :32-39, enums :382-412, Params :419-514,
// constructors :516-548, solve :567-577, getters :609-824, reset :830-908, getParams :914.
//
// With Eigen available (<Eigen/Core> found, or TEASER_HIP_USE_EIGEN defined) the matrix types ARE the
// reference's Eigen types, so existing call sites compile unchanged and src.data() is handed to the
// device zero-copy (Matrix<double,3,Dynamic> is column-major = the ABI's xyzxyz... layout).  Without
// Eigen (this repo's image has none) the same members are small value types with the accessors the
// reference's examples use (operator()(r,c), data(), cols()).
//
// Differences, all documented in DESIGN.md: the object is reusable (the reference object is
// single-use, registration.cc:702-704); `inlier_selection_mode` / `rotation_tim_graph` are honoured
// (the reference snapshot never stores params_); M-sized products (getSrcTIMs, getScaleInliersMask,
// ...) are not materialised on the device -- the scale-inlier set is available as the inlier graph.
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
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
#include "teaser_hip.h"

namespace teaser {

#if TEASER_HIP_HAVE_EIGEN
using Matrix3X = Eigen::Matrix<double, 3, Eigen::Dynamic>;
using Matrix3 = Eigen::Matrix3d;
using Vector3 = Eigen::Vector3d;
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
#endif

struct RegistrationSolution {  // registration.h:32-39
  bool valid = true;
  double scale;
  Vector3 translation;
  Matrix3 rotation;
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
    solution_ = RegistrationSolution();
    have_solution_ = false;
  }
  Params getParams() { return params_; }  // :914

  // registration.h:576-577 -- 3 x N matrices of corresponding points (column-major doubles)
  RegistrationSolution solve(const Matrix3X& src, const Matrix3X& dst) {
    if (src.cols() != dst.cols()) throw std::invalid_argument("solve: src and dst differ in size");
    teaser_solution_c o;
    check(teaser_hip_solve(h_, src.data(), dst.data(), (int32_t)src.cols(), &o));
    return adopt(o);
  }
  // registration.h:567-569 -- clouds + correspondences (src index, dst index)
  RegistrationSolution solve(const PointCloud& src_cloud, const PointCloud& dst_cloud,
                             const std::vector<std::pair<int, int>> correspondences) {
    static_assert(sizeof(PointXYZ) == 12 && sizeof(std::pair<int, int>) == 8, "packed layouts expected");
    teaser_solution_c o;
    check(teaser_hip_solve_correspondences(
        h_, reinterpret_cast<const float*>(src_cloud.data()), (int32_t)src_cloud.size(),
        reinterpret_cast<const float*>(dst_cloud.data()), (int32_t)dst_cloud.size(),
        reinterpret_cast<const int32_t*>(correspondences.data()), (int32_t)correspondences.size(), &o));
    return adopt(o);
  }

  // Stage entry points (registration.h:593-601): 3 x K TIMs / points -> the stage's estimate; the
  // result is also stored in the solution, like the reference does.
  Matrix3 solveForRotation(const Matrix3X& v1, const Matrix3X& v2) {
    if (v1.cols() != v2.cols()) throw std::invalid_argument("solveForRotation: sizes differ");
    double R[9], cost = 0;
    int32_t iters = 0;
    stage_mask_.assign((size_t)v1.cols(), 0);
    check(teaser_hip_solve_for_rotation(h_, v1.data(), v2.data(), (int32_t)v1.cols(), params_.noise_bound, R,
                                        stage_mask_.data(), &cost, &iters));
    raw_.gnc_cost = cost;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) solution_.rotation(r, c) = R[3 * r + c];
    return solution_.rotation;
  }
  Vector3 solveForTranslation(const Matrix3X& v1, const Matrix3X& v2) {
    if (v1.cols() != v2.cols()) throw std::invalid_argument("solveForTranslation: sizes differ");
    double t[3];
    stage_mask_.assign((size_t)v1.cols(), 0);
    check(teaser_hip_solve_for_translation(h_, v1.data(), v2.data(), (int32_t)v1.cols(), t, stage_mask_.data()));
    for (int r = 0; r < 3; ++r) solution_.translation(r) = t[r];
    return solution_.translation;
  }
  // inlier mask of the last stage call above (one entry per column)
  std::vector<bool> getStageInliersMask() const { return std::vector<bool>(stage_mask_.begin(), stage_mask_.end()); }

  RegistrationSolution getSolution() { return solution_; }                              // :617
  double getGNCRotationCostAtTermination() { return raw_.gnc_cost; }                    // :609-611
  std::vector<int> getInlierMaxClique() { return list(teaser_hip_get_max_clique); }     // :770
  std::vector<int> getRotationInliers() { return list(teaser_hip_get_rotation_inliers); }  // :713
  std::vector<int> getTranslationInliers() { return list(teaser_hip_get_translation_inliers); }  // :744
  std::vector<int> getInputOrderedTranslationInliers() {  // :752-763
    if (params_.rotation_estimation_algorithm == ROTATION_ESTIMATION_ALGORITHM::FGR)
      throw std::runtime_error(
          "This function is not supported when using FGR since FGR does not use max clique.");
    return list(teaser_hip_get_input_ordered_translation_inliers);
  }
  // masks over the clique ordering (registration.h:689, :723-725) and their maps = the clique (:698-737)
  std::vector<bool> getRotationInliersMask() { return mask(getRotationInliers(), n_rotation_tims()); }
  std::vector<bool> getTranslationInliersMask() { return mask(getTranslationInliers(), (size_t)raw_.clique_size); }
  std::vector<int> getRotationInliersMap() { return getInlierMaxClique(); }
  std::vector<int> getTranslationInliersMap() { return getInlierMaxClique(); }
  // registration.h:772: adjacency list of the inlier graph, unpacked from the device's bit matrix
  std::vector<std::vector<int>> getInlierGraph() {
    const int n = raw_.n, W = (n + 63) / 64;
    std::vector<std::vector<int>> adj((size_t)(n > 0 ? n : 0));
    if (!have_solution_ || n <= 0) return adj;
    int64_t len = 0;
    check(teaser_hip_get_inlier_graph_bitmap(h_, 0, nullptr, &len));
    std::vector<uint64_t> bm((size_t)len);
    if (len == 0) return adj;
    check(teaser_hip_get_inlier_graph_bitmap(h_, 0, bm.data(), &len));
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
  // registration.h:671-679: the TIMs (pairs of correspondences) that passed the scale stage
  std::vector<std::tuple<int, int>> getScaleInliers() {
    std::vector<std::tuple<int, int>> out;
    const auto adj = getInlierGraph();
    for (int i = 0; i < (int)adj.size(); ++i)
      for (int j : adj[(size_t)i])
        if (j > i) out.emplace_back(i, j);
    return out;
  }

  // not part of the reference surface: per-stage timings and the raw ABI record of the last solve
  teaser_hip_solver* handle() { return h_; }
  const teaser_solution_c& rawSolution() const { return raw_; }

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
  RegistrationSolution adopt(const teaser_solution_c& o) {
    raw_ = o;
    have_solution_ = true;
    solution_.valid = o.valid != 0;
    solution_.scale = o.scale;
    for (int r = 0; r < 3; ++r) {
      solution_.translation(r) = o.translation[r];
      for (int c = 0; c < 3; ++c) solution_.rotation(r, c) = o.rotation[3 * r + c];  // ABI: row-major
    }
    return solution_;
  }
  template <class F>
  std::vector<int> list(F getter) {
    std::vector<int> v;
    if (!have_solution_) return v;
    int64_t len = 0;
    check(getter(h_, 0, nullptr, &len));
    v.resize((size_t)len);
    if (len > 0) check(getter(h_, 0, reinterpret_cast<int32_t*>(v.data()), &len));
    return v;
  }
  static std::vector<bool> mask(const std::vector<int>& idx, size_t n) {
    std::vector<bool> m(n, false);
    for (int i : idx)
      if (i >= 0 && (size_t)i < n) m[(size_t)i] = true;
    return m;
  }
  size_t n_rotation_tims() const {  // CHAIN: K TIMs; COMPLETE: K (K - 1) / 2 (registration.cc:657-694)
    const size_t K = (size_t)raw_.clique_size;
    return params_.rotation_tim_graph == INLIER_GRAPH_FORMULATION::CHAIN ? K : K * (K - 1) / 2;
  }

  teaser_hip_solver* h_ = nullptr;
  Params params_;
  RegistrationSolution solution_;
  teaser_solution_c raw_{};
  bool have_solution_ = false;
  std::vector<uint8_t> stage_mask_;
};

}  // namespace teaser
