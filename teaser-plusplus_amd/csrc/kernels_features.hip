// kernels_features.hip -- the correspondence front-end on gfx950: FPFH descriptors and the mutual
// nearest-neighbour feature matcher, the stage BEFORE the registration hot path.
//
// Replaces, behind the C ABI (teaser_hip_compute_fpfh, teaser_hip_match_features):
//   * teaser::FPFHEstimation::computeFPFHFeatures (reference teaser/src/fpfh.cc:15-43), which is a
//     pass-through to PCL (NOT in the reference tree): pcl::NormalEstimation with a radius search
//     (features/normal_3d.hpp -> common/centroid.hpp computeMeanAndCovarianceMatrix, float accumulators ->
//     common/eigen.hpp eigen33, closed-form roots -> flipNormalTowardsViewpoint, viewpoint 0,0,0) and
//     pcl::FPFHEstimation (features/fpfh.hpp computePointSPFHSignature / weightPointSPFHSignature,
//     features/pfh_tools.hpp computePairFeatures), neighbours from pcl::search::KdTree::radiusSearch
//     (sorted by distance, squared distances, the query point included);
//   * the nearest-neighbour searches of teaser::Matcher::advancedMatching (reference
//     teaser/src/matcher.cc:117-192: FLANN KDTreeSingleIndex, exact L2 1-NN in 33 dimensions).
//
// Design.  Both PCL stages are "for every point, a SEQUENTIAL float reduction over its radius neighbours
// in order of increasing distance", and 71 000 histogram bins of the reference's fixture are decided by
// floor() of those floats -- so the order and the precision of every operation are part of the contract.
// The kd-trees are replaced by what a GPU is good at: brute-force O(n^2) distance tiles through LDS
// (count pass, exclusive scan, fill pass: exact neighbour lists, no tree, no traversal divergence), a
// per-point bitonic sort of (distance, index) in LDS, and one thread per point for the order-sensitive
// reductions (the same operation sequence as the CPU oracle: results are bit-identical to it).  libm's
// float functions are not portable to the last bit, so acos / atan2 / sin / cos are evaluated from IEEE
// double basic operations in a fixed order (the oracle uses the same formulas).  This file is compiled
// with -ffp-contract=off like the rest of the library: no product-add is fused by the compiler; the ONE place where the
// reference's fixtures need fused multiply-adds -- the covariance accumulators of the normals -- says so explicitly.
#include "internal.h"

namespace thip {

namespace {

struct Nbr {
  float d2;
  int32_t idx;
};

// ---- deterministic elementary functions (IEEE double basic operations only, fixed order) ---------
__device__ double fdet_atan_d(double x) {  // |x| <= 1: two argument halvings, then the Taylor series
  double t = x / (1.0 + __builtin_sqrt(1.0 + x * x));
  t = t / (1.0 + __builtin_sqrt(1.0 + t * t));
  const double t2 = t * t;
  double s = 0.0;
#pragma unroll  // (the quotients 1 / (2k + 1) fold to constants: correctly rounded at compile time as at run time)
  for (int k = 24; k >= 0; --k) s = 1.0 / (double)(2 * k + 1) - t2 * s;
  return 4.0 * (t * s);
}
__device__ double fdet_atan2_d(double y, double x) {
  const double pi = 3.14159265358979323846;
  if (x == 0.0 && y == 0.0) return 0.0;
  const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
  double a = (ax >= ay) ? fdet_atan_d(ay / ax) : pi / 2 - fdet_atan_d(ax / ay);
  if (x < 0.0) a = pi - a;
  return (y < 0.0) ? -a : a;
}
__device__ float fdet_atan2f(float y, float x) { return (float)fdet_atan2_d((double)y, (double)x); }
__device__ float fdet_acosf(float x) {  // x in [0, 1]
  const double xd = (double)x;
  const double s = __builtin_sqrt((1.0 - xd) * (1.0 + xd));
  return (float)fdet_atan2_d(s, xd);
}
__device__ void fdet_sincosf(float th, float* sn, float* cs) {
  const double t = (double)th, t2 = t * t;
  double s = 0.0, c = 0.0;
#pragma unroll 1
  for (int k = 12; k >= 1; --k) {
    s = 1.0 - t2 * s / (double)((2 * k) * (2 * k + 1));
    c = 1.0 - t2 * c / (double)((2 * k - 1) * (2 * k));
  }
  *sn = (float)(t * s);
  *cs = (float)c;
}

// squared distance exactly as flann::L2_Simple<float> accumulates it over x, y, z
__device__ __forceinline__ float feat_d2(float qx, float qy, float qz, float x, float y, float z) {
  const float dx = x - qx, dy = y - qy, dz = z - qz;
  float d2 = dx * dx;
  d2 += dy * dy;
  d2 += dz * dz;
  return d2;
}

// ---- radius neighbour lists: count pass / fill pass over LDS tiles of the cloud ------------------
// grid (blocks of 64 queries, chunks of kFeatChunk data points): a wave owns 64 queries and one chunk, so that a
// 5 000-point cloud is 800 waves (the first version's 20 workgroups left 236 of the 256 CUs idle: 0.39 ms for
// 25 M distance evaluations).  The lists are SORTED by (d2, idx) afterwards (feat_sort_kernel), so the chunks
// may append in any order: the count pass adds into counts[q], the fill pass reserves its run of the list with
// one atomic per (query, chunk).
constexpr int kFeatChunk = 512;

// FILL == 0: counts[q] += |{ i in chunk : d2(q, i) < r2 }| (counts zeroed by the launcher);
// FILL == 1: the chunk's neighbours (d2, i) into list[offset[q] + cursor[q] ...] (cursor zeroed by the launcher)
template <int FILL>
__global__ __launch_bounds__(64) void feat_radius_kernel(const float* __restrict__ pts, int n, float r2,
                                                         int32_t* __restrict__ counts,
                                                         const int64_t* __restrict__ offsets,
                                                         Nbr* __restrict__ list) {
  __shared__ float tile[kFeatChunk * 3];
  const int q = blockIdx.x * 64 + threadIdx.x;
  const bool live = q < n;
  const float qx = live ? pts[3 * q] : 0.f, qy = live ? pts[3 * q + 1] : 0.f, qz = live ? pts[3 * q + 2] : 0.f;
  const int base = blockIdx.y * kFeatChunk;
  const int m = min(kFeatChunk, n - base);
  for (int k = threadIdx.x; k < 3 * m; k += 64) tile[k] = pts[3 * base + k];
  __syncthreads();
  if (!live) return;
  int cnt = 0;
#pragma unroll 4
  for (int k = 0; k < m; ++k) cnt += feat_d2(qx, qy, qz, tile[3 * k], tile[3 * k + 1], tile[3 * k + 2]) < r2 ? 1 : 0;
  if (cnt == 0) return;
  const int pos = atomicAdd(&counts[q], cnt);
  if (!FILL) return;
  Nbr* out = list + offsets[q] + pos;
  int c = 0;
  for (int k = 0; k < m; ++k) {
    const float d2 = feat_d2(qx, qy, qz, tile[3 * k], tile[3 * k + 1], tile[3 * k + 2]);
    if (d2 < r2) {
      out[c].d2 = d2;
      out[c].idx = base + k;
      ++c;
    }
  }
}

// exclusive scan of the counts (one workgroup; n <= a few million) + the largest count
__global__ __launch_bounds__(1024) void feat_scan_kernel(const int32_t* __restrict__ counts, int n,
                                                         int64_t* __restrict__ offsets,
                                                         int64_t* __restrict__ total_and_max) {
  __shared__ long long part[1024];
  __shared__ int pmax[1024];
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int b = t * per, e = min(n, b + per);
  long long s = 0;
  int mx = 0;
  for (int i = b; i < e; ++i) {
    s += counts[i];
    mx = max(mx, counts[i]);
  }
  part[t] = s;
  pmax[t] = mx;
  __syncthreads();
  if (t == 0) {
    long long acc = 0;
    int m = 0;
    for (int k = 0; k < 1024; ++k) {
      const long long v = part[k];
      part[k] = acc;
      acc += v;
      m = max(m, pmax[k]);
    }
    total_and_max[0] = acc;
    total_and_max[1] = m;
  }
  __syncthreads();
  long long acc = part[t];
  for (int i = b; i < e; ++i) {
    offsets[i] = acc;
    acc += counts[i];
  }
  if (t == 1023) offsets[n] = part[1023] + s;
}

// per point: bitonic sort of its list by (d2, idx) in LDS (one 256-thread workgroup per point)
constexpr int kFeatSortCap = 4096;
__device__ __forceinline__ bool nbr_less(const Nbr& a, const Nbr& b) {
  return a.d2 < b.d2 || (a.d2 == b.d2 && a.idx < b.idx);
}
__global__ __launch_bounds__(256) void feat_sort_kernel(const int64_t* __restrict__ offsets,
                                                        const int32_t* __restrict__ counts,
                                                        Nbr* __restrict__ list) {
  __shared__ Nbr buf[kFeatSortCap];
  const int q = blockIdx.x;
  const int k = counts[q];
  if (k <= 1 || k > kFeatSortCap) return;  // (longer lists: feat_sort_long_kernel)
  Nbr* mine = list + offsets[q];
  int P2 = 2;
  while (P2 < k) P2 <<= 1;
  for (int i = threadIdx.x; i < P2; i += 256) {
    if (i < k) {
      buf[i] = mine[i];
    } else {
      buf[i].d2 = __builtin_inff();
      buf[i].idx = 0x7fffffff;
    }
  }
  __syncthreads();
  for (int size = 2; size <= P2; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = threadIdx.x; i < P2; i += 256) {
        const int l = i ^ stride;
        if (l > i) {
          const bool up = (i & size) == 0;
          const Nbr a = buf[i], b = buf[l];
          if (nbr_less(b, a) == up) {
            buf[i] = b;
            buf[l] = a;
          }
        }
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < k; i += 256) mine[i] = buf[i];
}

// Lists longer than the LDS sort holds (a radius that catches more than kFeatSortCap neighbours: dense clouds, large
// radii -- rare at the reference's settings, where a point has a few hundred): rank sort through a scratch copy, one
// workgroup per such list, O(k^2) comparisons.  (d2, idx) is a total order, so the result is the one the LDS sort
// gives; feat_sort_kernel skips these lists.
__global__ __launch_bounds__(256) void feat_sort_long_kernel(const int64_t* __restrict__ offsets,
                                                             const int32_t* __restrict__ counts, Nbr* __restrict__ list,
                                                             Nbr* __restrict__ scratch) {
  const int q = blockIdx.x;
  const int k = counts[q];
  if (k <= kFeatSortCap) return;
  Nbr* mine = list + offsets[q];
  Nbr* out = scratch + offsets[q];
  for (int i = threadIdx.x; i < k; i += 256) {
    const Nbr a = mine[i];
    int rank = 0;
    for (int j = 0; j < k; ++j) rank += nbr_less(mine[j], a) ? 1 : 0;
    out[rank] = a;
  }
  __threadfence_block();
  __syncthreads();
  for (int i = threadIdx.x; i < k; i += 256) mine[i] = out[i];
}

// ---- normals: pcl::NormalEstimation::computeFeature ----------------------------------------------
__device__ void feat_roots2(float b, float c, float* r) {  // pcl::computeRoots2
  r[0] = 0.0f;
  float d = b * b - 4.0f * c;
  if (d < 0.0f) d = 0.0f;
  const float sd = __builtin_sqrtf(d);
  r[2] = 0.5f * (b + sd);
  r[1] = 0.5f * (b - sd);
}
__device__ void feat_roots3(const float* m, float* r) {  // pcl::computeRoots, Scalar = float
  const float m00 = m[0], m01 = m[1], m02 = m[2], m11 = m[4], m12 = m[5], m22 = m[8];
  const float c0 = m00 * m11 * m22 + 2.0f * m01 * m02 * m12 - m00 * m12 * m12 - m11 * m02 * m02 - m22 * m01 * m01;
  const float c1 = m00 * m11 - m01 * m01 + m00 * m22 - m02 * m02 + m11 * m22 - m12 * m12;
  const float c2 = m00 + m11 + m22;
  if (__builtin_fabsf(c0) < 1.1920929e-07f) {
    feat_roots2(c2, c1, r);
    return;
  }
  const float s_inv3 = 1.0f / 3.0f, s_sqrt3 = __builtin_sqrtf(3.0f);
  const float c2_over_3 = c2 * s_inv3;
  float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.0f) a_over_3 = 0.0f;
  const float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
  float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.0f) q = 0.0f;
  const float rho = __builtin_sqrtf(-a_over_3);
  const float theta = fdet_atan2f(__builtin_sqrtf(-q), half_b) * s_inv3;
  float cos_theta, sin_theta;
  fdet_sincosf(theta, &sin_theta, &cos_theta);
  r[0] = c2_over_3 + 2.0f * rho * cos_theta;
  r[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  r[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  float t;
  if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; }
  if (r[1] >= r[2]) {
    t = r[1]; r[1] = r[2]; r[2] = t;
    if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; }
  }
  if (r[0] <= 0.0f) feat_roots2(c2, c1, r);
}
__device__ __forceinline__ void feat_cross(const float* a, const float* b, float* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

__global__ __launch_bounds__(128) void feat_normals_kernel(const float* __restrict__ pts, int n,
                                                           const int64_t* __restrict__ offsets,
                                                           const int32_t* __restrict__ counts,
                                                           const Nbr* __restrict__ list,
                                                           float* __restrict__ normals) {
  const int q = blockIdx.x * 128 + threadIdx.x;
  if (q >= n) return;
  const int k = counts[q];
  float* out = normals + 3 * q;
  if (k < 3) {  // computePointNormal: too few neighbours
    out[0] = out[1] = out[2] = __builtin_nanf("");
    return;
  }
  const Nbr* nb = list + offsets[q];
  // computeMeanAndCovarianceMatrix: nine float accumulators over the RAW coordinates, in list order,
  float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = 0; j < k; ++j) {
    const int i = nb[j].idx;
    const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    // FUSED multiply-adds, as an FMA-contracting (-march=native) build of the reference compiles PCL's
    // `accu[k] += p.a * p.b` -- the form the reference's fixtures were generated with (oracle/features_oracle.c,
    // feat_estimate_normals form 0: the matcher fixture's 189 pairs depend on it)
    acc[0] = __builtin_fmaf(x, x, acc[0]); acc[1] = __builtin_fmaf(x, y, acc[1]); acc[2] = __builtin_fmaf(x, z, acc[2]);
    acc[3] = __builtin_fmaf(y, y, acc[3]); acc[4] = __builtin_fmaf(y, z, acc[4]); acc[5] = __builtin_fmaf(z, z, acc[5]);
    acc[6] += x; acc[7] += y; acc[8] += z;
  }
  for (int i = 0; i < 9; ++i) acc[i] /= (float)k;
  float cov[9];
  cov[0] = __builtin_fmaf(-acc[6], acc[6], acc[0]);  // (`accu[k] - accu[6] * accu[6]`, fused likewise)
  cov[1] = __builtin_fmaf(-acc[6], acc[7], acc[1]);
  cov[2] = __builtin_fmaf(-acc[6], acc[8], acc[2]);
  cov[4] = __builtin_fmaf(-acc[7], acc[7], acc[3]);
  cov[5] = __builtin_fmaf(-acc[7], acc[8], acc[4]);
  cov[8] = __builtin_fmaf(-acc[8], acc[8], acc[5]);
  cov[3] = cov[1];
  cov[6] = cov[2];
  cov[7] = cov[5];
  // pcl::eigen33 (smallest eigenvalue and its eigenvector)
  float scale = 0.0f;
  for (int i = 0; i < 9; ++i) scale = __builtin_fmaxf(scale, __builtin_fabsf(cov[i]));
  if (scale <= 1.17549435e-38f) scale = 1.0f;
  float m[9];
  for (int i = 0; i < 9; ++i) m[i] = cov[i] / scale;
  float r[3];
  feat_roots3(m, r);
  m[0] -= r[0];
  m[4] -= r[0];
  m[8] -= r[0];
  float v1[3], v2[3], v3[3];
  feat_cross(m, m + 3, v1);
  feat_cross(m, m + 6, v2);
  feat_cross(m + 3, m + 6, v3);
  const float l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
  const float l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
  const float l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
  float nv[3], l;
  if (l1 >= l2 && l1 >= l3) { nv[0] = v1[0]; nv[1] = v1[1]; nv[2] = v1[2]; l = l1; }
  else if (l2 >= l1 && l2 >= l3) { nv[0] = v2[0]; nv[1] = v2[1]; nv[2] = v2[2]; l = l2; }
  else { nv[0] = v3[0]; nv[1] = v3[1]; nv[2] = v3[2]; l = l3; }
  const float s = __builtin_sqrtf(l);
  nv[0] /= s;
  nv[1] /= s;
  nv[2] /= s;
  // flipNormalTowardsViewpoint, viewpoint (0, 0, 0)
  const float vx = 0.0f - pts[3 * q], vy = 0.0f - pts[3 * q + 1], vz = 0.0f - pts[3 * q + 2];
  const float cos_theta = vx * nv[0] + vy * nv[1] + vz * nv[2];
  if (cos_theta < 0) {
    nv[0] *= -1;
    nv[1] *= -1;
    nv[2] *= -1;
  }
  out[0] = nv[0];
  out[1] = nv[1];
  out[2] = nv[2];
}

// ---- SPFH / FPFH -----------------------------------------------------------------------------------
// Eigen::Vector4f dot with a zero 4th component, in the order Eigen's SSE reduction adds the lane products
__device__ __forceinline__ float feat_dot4(const float* x, const float* y) {
  const float p0 = x[0] * y[0], p1 = x[1] * y[1], p2 = x[2] * y[2];
  return (p0 + p2) + (p1 + 0.0f);
}
__device__ bool feat_pair_features(const float* p1, const float* n1, const float* p2, const float* n2, float* f) {
  float dp[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  const float f4 = __builtin_sqrtf(feat_dot4(dp, dp));
  if (f4 == 0.0f) return false;
  float a[3] = {n1[0], n1[1], n1[2]}, b[3] = {n2[0], n2[1], n2[2]};
  const float angle1 = feat_dot4(a, dp) / f4;
  const float angle2 = feat_dot4(b, dp) / f4;
  float f3;
  if (fdet_acosf(__builtin_fabsf(angle1)) > fdet_acosf(__builtin_fabsf(angle2))) {  // switch p1 and p2
    for (int i = 0; i < 3; ++i) {
      a[i] = n2[i];
      b[i] = n1[i];
      dp[i] *= -1;
    }
    f3 = -angle2;
  } else {
    f3 = angle1;
  }
  float v[3];
  feat_cross(dp, a, v);
  const float vn = __builtin_sqrtf(feat_dot4(v, v));
  if (vn == 0.0f) return false;
  v[0] /= vn;
  v[1] /= vn;
  v[2] /= vn;
  float w[3];
  feat_cross(a, v, w);
  f[1] = feat_dot4(v, b);
  f[0] = fdet_atan2f(feat_dot4(w, b), feat_dot4(a, b));
  f[2] = f3;
  f[3] = f4;
  return true;
}
// static_cast<int>(std::floor(x)) with the x86 result for NaN (INT_MIN -> clamped to bin 0)
__device__ __forceinline__ int feat_bin(double x) {
  if (!(x == x)) return 0;
  const double fl = __builtin_floor(x);
  int hi = fl < 0.0 ? 0 : (fl >= 11.0 ? 10 : (int)fl);
  return hi;
}

// computePointSPFHSignature: one WAVE per point, its lanes over the neighbours.  Every contributing pair adds
// the SAME float (incr) to one bin of each of the three histograms, so a bin's value is incr added count times
// in float -- whatever the order of the neighbours: the lanes count into LDS, then lane b replays the additions
// of bin b.  (One thread per point, the first version, was 79 waves on 1 024 SIMDs and 1.98 ms at n = 5 000.)
__global__ __launch_bounds__(256) void feat_spfh_kernel(const float* __restrict__ pts,
                                                        const float* __restrict__ normals, int n,
                                                        const int64_t* __restrict__ offsets,
                                                        const int32_t* __restrict__ counts,
                                                        const Nbr* __restrict__ list, float* __restrict__ spfh) {
  __shared__ int bins[4][33];
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + w;
  if (lane < 33) bins[w][lane] = 0;
  __syncthreads();
  const int k = p < n ? counts[p] : 0;
  if (p < n) {
    const Nbr* nb = list + offsets[p];
    const float d_pi = 1.0f / (2.0f * 3.14159274101257324f);  // 1.0f / (2.0f * static_cast<float>(M_PI))
    const double pi = 3.14159265358979323846;
    const float P[3] = {pts[3 * p], pts[3 * p + 1], pts[3 * p + 2]};
    const float N[3] = {normals[3 * p], normals[3 * p + 1], normals[3 * p + 2]};
    for (int j = lane; j < k; j += 64) {
      const int qi = nb[j].idx;
      if (qi == p) continue;
      const float Q[3] = {pts[3 * qi], pts[3 * qi + 1], pts[3 * qi + 2]};
      const float M[3] = {normals[3 * qi], normals[3 * qi + 1], normals[3 * qi + 2]};
      float f[4];
      if (!feat_pair_features(P, N, Q, M, f)) continue;
      atomicAdd(&bins[w][feat_bin(11 * (((double)f[0] + pi) * (double)d_pi))], 1);
      atomicAdd(&bins[w][11 + feat_bin(11 * (((double)f[1] + 1.0) * 0.5))], 1);
      atomicAdd(&bins[w][22 + feat_bin(11 * (((double)f[2] + 1.0) * 0.5))], 1);
    }
  }
  __syncthreads();
  if (p < n && lane < 33) {
    const float incr = 100.0f / (float)(k - 1);
    const int c = bins[w][lane];
    float h = 0.0f;
    for (int i = 0; i < c; ++i) h += incr;
    spfh[(size_t)p * 33 + lane] = h;
  }
}

// weightPointSPFHSignature: one wave per point, lane b = bin b; the neighbours in order of increasing distance
// (the float sums depend on it).  Per neighbour the bin values val_b = spfh[b] * weight are one coalesced row;
// the group sums add them in the reference's order (j outer, bin inner), every lane of a group carrying its
// group's running sum and fetching the eleven addends across the lanes.  A skipped neighbour (d2 == 0)
// contributes +0, which leaves the non-negative sums unchanged bit for bit.  Neighbour records are wave-uniform
// (scalar loads), fetched one block of kFpfhBlock ahead of the rows they index.
constexpr int kFpfhBlock = 8;
__global__ __launch_bounds__(256) void feat_fpfh_kernel(int n, const int64_t* __restrict__ offsets,
                                                        const int32_t* __restrict__ counts,
                                                        const Nbr* __restrict__ list,
                                                        const float* __restrict__ spfh, float* __restrict__ out) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + w;
  if (p >= n) return;
  const int k = counts[p];
  const Nbr* nb = list + offsets[p];
  const int b = lane < 33 ? lane : 32;  // (lanes 33 .. 63 shadow bin 32; they do not store)
  const int g0 = 11 * (b / 11);
  float o = 0.0f, sum = 0.0f;
  Nbr cur[kFpfhBlock], nxt[kFpfhBlock];
#pragma unroll
  for (int u = 0; u < kFpfhBlock; ++u) {
    cur[u].d2 = u < k ? nb[u].d2 : 0.0f;
    cur[u].idx = u < k ? nb[u].idx : p;
  }
  for (int j0 = 0; j0 < k; j0 += kFpfhBlock) {
    float hv[kFpfhBlock];
#pragma unroll
    for (int u = 0; u < kFpfhBlock; ++u) hv[u] = spfh[(size_t)cur[u].idx * 33 + b];
#pragma unroll
    for (int u = 0; u < kFpfhBlock; ++u) {
      const int j = j0 + kFpfhBlock + u;
      nxt[u].d2 = j < k ? nb[j].d2 : 0.0f;
      nxt[u].idx = j < k ? nb[j].idx : p;
    }
#pragma unroll
    for (int u = 0; u < kFpfhBlock; ++u) {
      const float d2 = cur[u].d2;
      const float weight = 1.0f / d2;
      const float val = d2 == 0.0f ? 0.0f : hv[u] * weight;
      o += val;
#pragma unroll
      for (int i = 0; i < 11; ++i) sum += __shfl(val, g0 + i);
    }
#pragma unroll
    for (int u = 0; u < kFpfhBlock; ++u) cur[u] = nxt[u];
  }
  if (sum != 0) sum = 100.0f / sum;
  if (lane < 33) out[(size_t)p * 33 + lane] = o * sum;
}

// ---- exact L2 1-NN in `dim` dimensions (the matcher's two searches) -------------------------------
// grid (query blocks of 64, data chunks of kNnChunk): a thread owns one query (its `dim` values staged in
// LDS, one row per thread), walks the chunk's data points through an LDS tile, accumulating
// (q_c - d_c)^2 over c in order in float; first strict minimum = lowest index on ties.
// Exact L2 1-NN, brute force.  The arithmetic of one distance is fixed by the oracle (flann::L2<float> order:
// d += (q_c - x_c)^2 for c = 0 .. dim-1, float), so the speed has to come from the layout: the query vector of a
// thread lives in REGISTERS (dim is a template parameter for the 33-bin FPFH signature; other dims take the generic
// instantiation with the query in LDS), a tile of kNnTile data points is staged in LDS and read back as broadcast
// ds_read_b128, and the data set is cut into chunks of kNnChunk points so that a 5 000 x 5 000 match fills the GPU
// (1 700 waves instead of the 164 of the first version, which took 6 ms per direction -- 0.3 TFLOP/s).
constexpr int kNnChunk = 256;
constexpr int kNnTile = 64;
constexpr int kNnMaxDim = 64;
template <int DIM>
__global__ __launch_bounds__(64) void feat_nn_partial_kernel(const float* __restrict__ data, int nd,
                                                             const float* __restrict__ query, int nq, int dim_rt,
                                                             float* __restrict__ part_d,
                                                             int32_t* __restrict__ part_i) {
  constexpr int kPad = DIM > 0 ? ((DIM + 3) & ~3) : kNnMaxDim;  // floats per staged point (16-byte rows)
  __shared__ __attribute__((aligned(16))) float tile[kNnTile * kPad];
  __shared__ float qs[DIM > 0 ? 1 : 64 * (kNnMaxDim + 1)];
  const int dim = DIM > 0 ? DIM : dim_rt;
  const int q = blockIdx.x * 64 + threadIdx.x;
  const bool live = q < nq;
  float qr[DIM > 0 ? kPad : 1];
  if (DIM > 0) {
#pragma unroll
    for (int c = 0; c < kPad; ++c) qr[c] = (live && c < DIM) ? query[(size_t)q * DIM + c] : 0.f;
  } else {
    for (int c = 0; c < dim; ++c) qs[threadIdx.x * (kNnMaxDim + 1) + c] = live ? query[(size_t)q * dim + c] : 0.f;
  }
  const int lo = blockIdx.y * kNnChunk, hi = min(nd, lo + kNnChunk);
  float best = __builtin_inff();
  int bi = -1;
  for (int base = lo; base < hi; base += kNnTile) {
    const int m = min(kNnTile, hi - base);
    __syncthreads();
    if (DIM > 0) {
      for (int k = threadIdx.x; k < m * kPad; k += 64) {
        const int pt = k / kPad, c = k - pt * kPad;
        tile[k] = c < DIM ? data[(size_t)(base + pt) * DIM + c] : 0.f;
      }
    } else {
      for (int k = threadIdx.x; k < m * dim; k += 64) tile[(k / dim) * kPad + (k % dim)] = data[(size_t)base * dim + k];
    }
    __syncthreads();
    for (int k = 0; k < m; ++k) {
      float d = 0;
      if (DIM > 0) {
        const float4* row = reinterpret_cast<const float4*>(tile + k * kPad);
#pragma unroll
        for (int c4 = 0; c4 < kPad / 4; ++c4) {
          const float4 x = row[c4];  // broadcast: every lane reads the same 16 bytes
          const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (4 * c4 + e < DIM) {  // (the padding is skipped: d + 0 * 0 would still be a float operation)
              const float t = qr[4 * c4 + e] - xs[e];
              d += t * t;
            }
        }
      } else {
        for (int c = 0; c < dim; ++c) {
          const float t = qs[threadIdx.x * (kNnMaxDim + 1) + c] - tile[k * kPad + c];
          d += t * t;
        }
      }
      if (d < best) {
        best = d;
        bi = base + k;
      }
    }
  }
  if (live) {
    part_d[(size_t)blockIdx.y * nq + q] = best;
    part_i[(size_t)blockIdx.y * nq + q] = bi;
  }
}
__global__ __launch_bounds__(256) void feat_nn_final_kernel(const float* __restrict__ part_d,
                                                            const int32_t* __restrict__ part_i, int nq,
                                                            int chunks, int32_t* __restrict__ nn) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  float best = __builtin_inff();
  int bi = -1;
  for (int c = 0; c < chunks; ++c) {  // chunks in index order + strict <: the first minimum overall
    const float d = part_d[(size_t)c * nq + q];
    if (d < best) {
      best = d;
      bi = part_i[(size_t)c * nq + q];
    }
  }
  nn[q] = bi;
}

}  // namespace

// ---- launchers ---------------------------------------------------------------------------------------
int64_t feat_nbr_bytes() { return (int64_t)sizeof(Nbr); }

void launch_feat_radius_count(hipStream_t s, const float* d_pts, int n, float r2, int32_t* d_counts) {
  if (n <= 0) return;
  (void)hipMemsetAsync(d_counts, 0, (size_t)n * 4, s);
  hipLaunchKernelGGL(feat_radius_kernel<0>, dim3((n + 63) / 64, (n + kFeatChunk - 1) / kFeatChunk), dim3(64), 0, s, d_pts,
                     n, r2, d_counts, static_cast<const int64_t*>(nullptr), static_cast<Nbr*>(nullptr));
}
void launch_feat_scan(hipStream_t s, const int32_t* d_counts, int n, int64_t* d_offsets, int64_t* d_total_max) {
  hipLaunchKernelGGL(feat_scan_kernel, dim3(1), dim3(1024), 0, s, d_counts, n, d_offsets, d_total_max);
}
void launch_feat_radius_fill_sort(hipStream_t s, const float* d_pts, int n, float r2, const int32_t* d_counts,
                                  int32_t* d_cursor, const int64_t* d_offsets, void* d_list) {
  if (n <= 0) return;
  (void)hipMemsetAsync(d_cursor, 0, (size_t)n * 4, s);
  hipLaunchKernelGGL(feat_radius_kernel<1>, dim3((n + 63) / 64, (n + kFeatChunk - 1) / kFeatChunk), dim3(64), 0, s, d_pts,
                     n, r2, d_cursor, d_offsets, reinterpret_cast<Nbr*>(d_list));
  hipLaunchKernelGGL(feat_sort_kernel, dim3(n), dim3(256), 0, s, d_offsets, d_counts, reinterpret_cast<Nbr*>(d_list));
}
int feat_sort_capacity() { return kFeatSortCap; }
void launch_feat_sort_long(hipStream_t s, int n, const int32_t* d_counts, const int64_t* d_offsets, void* d_list,
                           void* d_scratch) {
  if (n <= 0) return;
  hipLaunchKernelGGL(feat_sort_long_kernel, dim3(n), dim3(256), 0, s, d_offsets, d_counts, reinterpret_cast<Nbr*>(d_list),
                     reinterpret_cast<Nbr*>(d_scratch));
}
void launch_feat_normals(hipStream_t s, const float* d_pts, int n, const int64_t* d_offsets, const int32_t* d_counts,
                         const void* d_list, float* d_normals) {
  if (n <= 0) return;
  hipLaunchKernelGGL(feat_normals_kernel, dim3((n + 127) / 128), dim3(128), 0, s, d_pts, n, d_offsets, d_counts,
                     reinterpret_cast<const Nbr*>(d_list), d_normals);
}
void launch_feat_fpfh(hipStream_t s, const float* d_pts, const float* d_normals, int n, const int64_t* d_offsets,
                      const int32_t* d_counts, const void* d_list, float* d_spfh, float* d_out) {
  if (n <= 0) return;
  hipLaunchKernelGGL(feat_spfh_kernel, dim3((n + 3) / 4), dim3(256), 0, s, d_pts, d_normals, n, d_offsets, d_counts,
                     reinterpret_cast<const Nbr*>(d_list), d_spfh);
  hipLaunchKernelGGL(feat_fpfh_kernel, dim3((n + 3) / 4), dim3(256), 0, s, n, d_offsets, d_counts,
                     reinterpret_cast<const Nbr*>(d_list), d_spfh, d_out);
}
int feat_nn_chunks(int nd) { return (nd + kNnChunk - 1) / kNnChunk; }
int feat_nn_max_dim() { return kNnMaxDim; }
void launch_feat_nn1(hipStream_t s, const float* d_data, int nd, const float* d_query, int nq, int dim,
                     float* d_part_d, int32_t* d_part_i, int32_t* d_nn) {
  if (nq <= 0 || nd <= 0) return;
  const int chunks = feat_nn_chunks(nd);
  if (dim == 33)  // pcl::FPFHSignature33
    hipLaunchKernelGGL(feat_nn_partial_kernel<33>, dim3((nq + 63) / 64, chunks), dim3(64), 0, s, d_data, nd, d_query, nq,
                       dim, d_part_d, d_part_i);
  else
    hipLaunchKernelGGL(feat_nn_partial_kernel<0>, dim3((nq + 63) / 64, chunks), dim3(64), 0, s, d_data, nd, d_query, nq,
                       dim, d_part_d, d_part_i);
  hipLaunchKernelGGL(feat_nn_final_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, d_part_d, d_part_i, nq, chunks,
                     d_nn);
}

}  // namespace thip
