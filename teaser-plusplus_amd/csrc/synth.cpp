// synth.cpp -- deterministic synthetic registration problems (host only, no GPU needed).
//
// The reference has no seeded generator (its tests/examples use std::random_device:
// reference test/teaser/registration-test.cc:398-431, examples/teaser_cpp_ply/teaser_cpp_ply.cc:
// 21-40), so the benchmark inputs are defined here:
//   splitmix64(seed) -> doubles (x>>11)*2^-53;  src ~ U[0,1)^3;  R from a normalised Gaussian
//   quaternion (Box-Muller on the same stream);  t ~ U[-1,1)^3;
//   inliers  dst = R src + t + eps, eps ~ U[-nb/sqrt3, nb/sqrt3]^3 (so |eps| <= nb and every
//            inlier pair passes the 2*nb TIM test: the inliers form a clique);
//   outliers exactly round(rho*N) distinct indices (seeded Fisher-Yates prefix),
//            dst ~ U(ball(centre R*(1/2,1/2,1/2)+t, radius sqrt3/2)).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <ctime>
#include <utility>
#include <vector>

#include "teaser_hip.h"

namespace {
struct SplitMix64 {
  uint64_t s;
  explicit SplitMix64(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};
}  // namespace

extern "C" int32_t teaser_hip_synth_problem(uint64_t seed, int32_t n, double outlier_ratio,
                                            double noise_bound, double* src, double* dst,
                                            double* R_out, double* t_out, uint8_t* inlier_mask) {
  if (n < 0 || !src || !dst || outlier_ratio < 0 || outlier_ratio > 1) return TEASER_HIP_ERR_BAD_ARG;
  SplitMix64 rng(seed);
  // rotation from a normalised 4-vector of N(0,1)
  double q[4];
  for (int k = 0; k < 2; ++k) {
    double u1 = rng.uniform(), u2 = rng.uniform();
    if (u1 < 1e-300) u1 = 1e-300;
    const double r = std::sqrt(-2.0 * std::log(u1));
    q[2 * k] = r * std::cos(6.283185307179586476925286766559 * u2);
    q[2 * k + 1] = r * std::sin(6.283185307179586476925286766559 * u2);
  }
  double qn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (qn < 1e-300) {
    q[0] = 1;
    q[1] = q[2] = q[3] = 0;
    qn = 1;
  }
  const double w = q[0] / qn, x = q[1] / qn, y = q[2] / qn, z = q[3] / qn;
  double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w),     2 * (x * z + y * w),
                 2 * (x * y + z * w),     1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                 2 * (x * z - y * w),     2 * (y * z + x * w),     1 - 2 * (x * x + y * y)};
  double t[3];
  for (int k = 0; k < 3; ++k) t[k] = 2.0 * rng.uniform() - 1.0;
  for (int64_t i = 0; i < 3 * (int64_t)n; ++i) src[i] = rng.uniform();
  const double a = noise_bound / std::sqrt(3.0);
  for (int64_t i = 0; i < n; ++i) {
    const double* s = src + 3 * i;
    for (int r = 0; r < 3; ++r) {
      const double eps = (2.0 * rng.uniform() - 1.0) * a;
      dst[3 * i + r] = (R[3 * r] * s[0] + R[3 * r + 1] * s[1] + R[3 * r + 2] * s[2]) + t[r] + eps;
    }
  }
  int64_t n_out = (int64_t)std::llround(outlier_ratio * (double)n);
  if (n_out > n) n_out = n;
  std::vector<int32_t> perm((size_t)n);
  for (int32_t i = 0; i < n; ++i) perm[(size_t)i] = i;
  if (inlier_mask)
    for (int32_t i = 0; i < n; ++i) inlier_mask[i] = 1;
  double centre[3];
  for (int r = 0; r < 3; ++r) centre[r] = 0.5 * (R[3 * r] + R[3 * r + 1] + R[3 * r + 2]) + t[r];
  const double radius = std::sqrt(3.0) / 2.0;
  for (int64_t k = 0; k < n_out; ++k) {
    int64_t j = k + (int64_t)(rng.uniform() * (double)(n - k));
    if (j >= n) j = n - 1;
    const int32_t tmp = perm[(size_t)k];
    perm[(size_t)k] = perm[(size_t)j];
    perm[(size_t)j] = tmp;
    const int64_t v = perm[(size_t)k];
    double p[3];
    while (true) {
      double nn = 0;
      for (int r = 0; r < 3; ++r) {
        p[r] = 2.0 * rng.uniform() - 1.0;
        nn += p[r] * p[r];
      }
      if (nn <= 1.0) break;
    }
    for (int r = 0; r < 3; ++r) dst[3 * v + r] = centre[r] + radius * p[r];
    if (inlier_mask) inlier_mask[v] = 0;
  }
  if (R_out)
    for (int k = 0; k < 9; ++k) R_out[k] = R[k];
  if (t_out)
    for (int k = 0; k < 3; ++k) t_out[k] = t[k];
  return TEASER_HIP_OK;
}

// The tuple constraint of teaser::Matcher::advancedMatching (reference teaser/src/matcher.cc:223-283), host
// arithmetic in float as there.  seed = 0: seeded from the clock (the reference: srand(time(NULL))).
extern "C" int32_t teaser_hip_tuple_test(teaser_hip_solver*, const float* src_xyz, int32_t n_src, const float* dst_xyz,
                                         int32_t n_dst, float tuple_scale, uint64_t seed, int32_t* pairs,
                                         int64_t* n_pairs) {
  if (!n_pairs || *n_pairs < 0 || n_src < 0 || n_dst < 0 || (*n_pairs > 0 && (!pairs || !src_xyz || !dst_xyz)))
    return TEASER_HIP_ERR_BAD_ARG;
  const int64_t ncorr = *n_pairs;
  if (!(tuple_scale > 0.0f) || ncorr == 0) return TEASER_HIP_OK;  // matcher.cc:223: skipped for tuple_scale == 0
  for (int64_t k = 0; k < ncorr; ++k)
    if (pairs[2 * k] < 0 || pairs[2 * k] >= n_src || pairs[2 * k + 1] < 0 || pairs[2 * k + 1] >= n_dst)
      return TEASER_HIP_ERR_BAD_ARG;
  SplitMix64 rng(seed ? seed : (uint64_t)time(nullptr));
  auto dist = [](const float* p, int a, int b) {
    const float dx = p[3 * a] - p[3 * b], dy = p[3 * a + 1] - p[3 * b + 1], dz = p[3 * a + 2] - p[3 * b + 2];
    return std::sqrt(dx * dx + dy * dy + dz * dz);
  };
  const float scale = tuple_scale;
  const int64_t trials = ncorr * 100;  // matcher.cc:231
  std::vector<std::pair<int32_t, int32_t>> kept;
  for (int64_t i = 0; i < trials; ++i) {
    const int64_t r0 = (int64_t)(rng.next() % (uint64_t)ncorr), r1 = (int64_t)(rng.next() % (uint64_t)ncorr),
                  r2 = (int64_t)(rng.next() % (uint64_t)ncorr);
    const int i0 = pairs[2 * r0], j0 = pairs[2 * r0 + 1], i1 = pairs[2 * r1], j1 = pairs[2 * r1 + 1], i2 = pairs[2 * r2],
              j2 = pairs[2 * r2 + 1];
    const float li0 = dist(src_xyz, i0, i1), li1 = dist(src_xyz, i1, i2), li2 = dist(src_xyz, i2, i0);
    const float lj0 = dist(dst_xyz, j0, j1), lj1 = dist(dst_xyz, j1, j2), lj2 = dist(dst_xyz, j2, j0);
    if ((li0 * scale < lj0) && (lj0 < li0 / scale) && (li1 * scale < lj1) && (lj1 < li1 / scale) && (li2 * scale < lj2) &&
        (lj2 < li2 / scale)) {  // matcher.cc:267-268
      kept.emplace_back(i0, j0);
      kept.emplace_back(i1, j1);
      kept.emplace_back(i2, j2);
    }
  }
  std::sort(kept.begin(), kept.end());  // matcher.cc:299-300 (every kept pair is one of the input pairs: it fits)
  kept.erase(std::unique(kept.begin(), kept.end()), kept.end());
  for (size_t k = 0; k < kept.size(); ++k) {
    pairs[2 * k] = kept[k].first;
    pairs[2 * k + 1] = kept[k].second;
  }
  *n_pairs = (int64_t)kept.size();
  return TEASER_HIP_OK;
}
