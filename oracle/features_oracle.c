/*
 * features_oracle.c -- CPU restatement of the correspondence front-end that sits BEFORE the
 * registration hot path: FPFH descriptors (reference teaser/src/fpfh.cc:15-43, a pass-through to PCL's
 * NormalEstimation + FPFHEstimation) and the feature matcher (reference teaser/src/matcher.cc:21-301,
 * FLANN 1-NN both ways + cross check + optional tuple test).
 *
 * TEST INFRASTRUCTURE ONLY (same rules as teaser_oracle.c): nothing in the product links or calls it.
 *
 * The arithmetic lives in third-party libraries that are absent from /root/reference:
 *   - PCL (FindPCL, teaser/CMakeLists.txt:82-97, version not pinned; "PCL 1.9" in the docs):
 *     pcl::NormalEstimation (features/normal_3d.hpp: computePointNormal -> computeMeanAndCovarianceMatrix,
 *     common/centroid.hpp, float accumulators -> solvePlaneParameters -> pcl::eigen33, common/eigen.hpp,
 *     closed-form roots, float -> flipNormalTowardsViewpoint, viewpoint (0,0,0)),
 *     pcl::FPFHEstimation (features/fpfh.hpp: computePointSPFHSignature, weightPointSPFHSignature;
 *     pcl::computePairFeatures, features/pfh_tools.hpp), pcl::search::KdTree::radiusSearch (sorted by
 *     distance, squared distances, the query point included);
 *   - FLANN (KDTreeSingleIndex, exact L2 1-NN).
 * Their published algorithms are restated here in the same precision (float); parity is PINNED on the
 * reference's own fixtures for this path: test/teaser/data/bunny.pcd -> bunny_fpfh.csv (feature-test.cc,
 * tolerance 1e-4) and matcher-test-object-1.ply / -scene-1.ply -> matcher-test-matches-1.csv
 * (matcher-test.cc:46-85), see tests/test_features_oracle.py.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FEAT_API __attribute__((visibility("default")))

/* ---- deterministic elementary functions -------------------------------------------------------------
 * PCL calls libm's float functions (atan2f / cosf / sinf in the eigen solver, acosf / atan2f in the pair
 * features); their last-bit behaviour differs between libm builds, and the HIP device library differs
 * again.  The restatement therefore evaluates them from IEEE basic operations only (+ - * / sqrt in
 * double, fixed evaluation order, -ffp-contract=off), rounded to float at the end: the same bits on every
 * host and -- with the same formulas -- on the GPU.  Accuracy ~1e-15 relative before the final rounding,
 * i.e. the correctly rounded float result except in astronomically rare cases. */
static double det_atan_d(double x) { /* |x| <= 1: argument halving twice, then a Taylor series */
  /* atan(x) = 2 atan(x / (1 + sqrt(1 + x^2))) */
  double t = x / (1.0 + sqrt(1.0 + x * x));
  t = t / (1.0 + sqrt(1.0 + t * t)); /* |t| <= tan(pi/16) ~ 0.199 */
  const double t2 = t * t;
  double s = 0.0;
  for (int k = 24; k >= 0; --k) s = 1.0 / (double)(2 * k + 1) - t2 * s; /* sum (-1)^k t^2k / (2k+1) */
  return 4.0 * (t * s);
}
static double det_atan2_d(double y, double x) {
  const double pi = 3.14159265358979323846;
  if (x == 0.0 && y == 0.0) return 0.0;
  const double ax = fabs(x), ay = fabs(y);
  double a = (ax >= ay) ? det_atan_d(ay / ax) : pi / 2 - det_atan_d(ax / ay); /* first octant pair */
  if (x < 0.0) a = pi - a;
  return (y < 0.0) ? -a : a;
}
static float det_atan2f(float y, float x) { return (float)det_atan2_d((double)y, (double)x); }
static float det_acosf(float x) { /* x in [0, 1]: acos(x) = atan2(sqrt(1 - x^2), x) */
  const double xd = (double)x;
  const double s = sqrt((1.0 - xd) * (1.0 + xd));
  return (float)det_atan2_d(s, xd);
}
static void det_sincosf(float th, float* sn, float* cs) { /* |th| <= pi/3 + (the eigen solver's range) */
  const double t = (double)th, t2 = t * t;
  double s = 0.0, c = 0.0;
  for (int k = 12; k >= 1; --k) { /* Horner on the Taylor series */
    s = 1.0 - t2 * s / (double)((2 * k) * (2 * k + 1));
    c = 1.0 - t2 * c / (double)((2 * k - 1) * (2 * k));
  }
  *sn = (float)(t * s);
  *cs = (float)c;
}

typedef struct {
  float d2;
  int32_t idx;
} nbr_t;

static int cmp_nbr(const void* a, const void* b) {
  const nbr_t* x = (const nbr_t*)a;
  const nbr_t* y = (const nbr_t*)b;
  if (x->d2 < y->d2) return -1;
  if (x->d2 > y->d2) return 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}

/* pcl::search::KdTree::radiusSearch (FLANN, L2_Simple<float>): squared distances accumulated in float over
 * x, y, z; neighbours with d2 < r^2... FLANN's RadiusResultSet keeps dist <= radius?  KDTreeSingleIndex
 * prunes with `worst_dist` = radius^2 and RadiusResultSet::addPoint keeps `dist < radius`; points exactly on
 * the sphere are measure-zero for the fixtures.  Sorted ascending by distance (sorted_results_ = true). */
static int radius_search(const float* pts, int n, int q, double radius, nbr_t* out) {
  const float r2 = (float)(radius * radius); /* pcl::KdTreeFLANN::radiusSearch: static_cast<float>(radius * radius) */
  const float qx = pts[3 * q], qy = pts[3 * q + 1], qz = pts[3 * q + 2];
  int k = 0;
  for (int i = 0; i < n; ++i) {
    const float dx = pts[3 * i] - qx, dy = pts[3 * i + 1] - qy, dz = pts[3 * i + 2] - qz;
    float d2 = dx * dx;
    d2 += dy * dy;
    d2 += dz * dz;
    if (d2 < r2) {
      out[k].d2 = d2;
      out[k].idx = i;
      ++k;
    }
  }
  qsort(out, (size_t)k, sizeof(nbr_t), cmp_nbr);
  return k;
}

/* pcl::computeRoots2 / computeRoots (common/eigen.hpp), Scalar = float */
static void roots2(float b, float c, float* r) {
  r[0] = 0.0f;
  float d = b * b - 4.0f * c;
  if (d < 0.0f) d = 0.0f;
  const float sd = sqrtf(d);
  r[2] = 0.5f * (b + sd);
  r[1] = 0.5f * (b - sd);
}
static void roots3(const float* m /* row-major 3x3 */, float* r) {
  const float m00 = m[0], m01 = m[1], m02 = m[2], m11 = m[4], m12 = m[5], m22 = m[8];
  const float c0 = m00 * m11 * m22 + 2.0f * m01 * m02 * m12 - m00 * m12 * m12 - m11 * m02 * m02 - m22 * m01 * m01;
  const float c1 = m00 * m11 - m01 * m01 + m00 * m22 - m02 * m02 + m11 * m22 - m12 * m12;
  const float c2 = m00 + m11 + m22;
  if (fabsf(c0) < 1.1920929e-07f) { /* std::numeric_limits<float>::epsilon(): one root is zero */
    roots2(c2, c1, r);
    return;
  }
  const float s_inv3 = 1.0f / 3.0f, s_sqrt3 = sqrtf(3.0f);
  const float c2_over_3 = c2 * s_inv3;
  float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.0f) a_over_3 = 0.0f;
  const float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
  float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.0f) q = 0.0f;
  const float rho = sqrtf(-a_over_3);
  const float theta = det_atan2f(sqrtf(-q), half_b) * s_inv3;
  float cos_theta, sin_theta;
  det_sincosf(theta, &sin_theta, &cos_theta);
  r[0] = c2_over_3 + 2.0f * rho * cos_theta;
  r[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  r[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  float t;
  if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; }
  if (r[1] >= r[2]) {
    t = r[1]; r[1] = r[2]; r[2] = t;
    if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; }
  }
  if (r[0] <= 0.0f) roots2(c2, c1, r); /* PSD matrix: a non-positive smallest root means it is zero */
}

static void cross3(const float* a, const float* b, float* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

/* pcl::eigen33(mat, eigenvalue, eigenvector): smallest eigenvalue and its eigenvector */
static void eigen33_smallest(const float* cov, float* eval, float* evec) {
  float scale = 0.0f;
  for (int i = 0; i < 9; ++i) scale = fmaxf(scale, fabsf(cov[i]));
  if (scale <= 1.17549435e-38f) scale = 1.0f; /* std::numeric_limits<float>::min() */
  float m[9];
  for (int i = 0; i < 9; ++i) m[i] = cov[i] / scale;
  float r[3];
  roots3(m, r);
  *eval = r[0] * scale;
  m[0] -= r[0];
  m[4] -= r[0];
  m[8] -= r[0];
  float v1[3], v2[3], v3[3];
  cross3(m, m + 3, v1);
  cross3(m, m + 6, v2);
  cross3(m + 3, m + 6, v3);
  const float l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
  const float l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
  const float l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
  const float* v;
  float l;
  if (l1 >= l2 && l1 >= l3) { v = v1; l = l1; }
  else if (l2 >= l1 && l2 >= l3) { v = v2; l = l2; }
  else { v = v3; l = l3; }
  const float s = sqrtf(l);
  evec[0] = v[0] / s;
  evec[1] = v[1] / s;
  evec[2] = v[2] / s;
}

/* pcl::NormalEstimation::computeFeature with setRadiusSearch(radius), viewpoint (0, 0, 0).
 * form 0 (the default of the Python wrapper): pcl::computeMeanAndCovarianceMatrix on the raw coordinates
 *   (common/centroid.hpp: `accu[k] += p.a * p.b` ... `cov = accu[k] - accu[6] * accu[6]`) AS A BUILD WITH FMA
 *   CONTRACTION COMPILES IT -- the reference's CMake turns -march=native on when PCL was built with it
 *   (CMakeLists.txt:27, 62-71), and GCC then fuses exactly these multiply-adds.  The covariance of a 2 cm
 *   neighbourhood a metre from the origin is the difference of numbers 10^4 times larger, so whether the products are
 *   rounded before they are added moves the normals by ~1e-3 rad -- and decides the reference's fixtures: with the
 *   fused form the matcher fixture (matcher-test.cc:46-85) is reproduced EXACTLY, all 189 pairs in order, and the
 *   bunny FPFH fixture agrees on all 13 101 bins with ONE forced near-tie; with every operation rounded (form 2) 174
 *   pairs and six forced near-ties (tests/test_features_oracle.py).  Nothing else in this file is contracted: the
 *   fused covariance alone reproduces what a whole-file -mfma -ffp-contract=fast build of this restatement gives.
 * form 1: PCL >= 1.10's computeMeanAndCovarianceMatrix (accumulates relative to the first neighbour), not fused.
 * form 2: form 0 without fused operations (a generic x86-64 build).
 * normals: n x 3 floats (NaN when < 3 neighbours). */
FEAT_API int feat_estimate_normals(const float* pts, int32_t n, double radius, int32_t form, float* normals) {
  const int centred = form == 1, fused = form == 0;
#pragma omp parallel
  {
    nbr_t* nb = (nbr_t*)malloc((size_t)n * sizeof(nbr_t));
#pragma omp for schedule(dynamic, 16)
    for (int q = 0; q < n; ++q) {
      const int k = radius_search(pts, n, q, radius, nb);
      float* out = normals + 3 * q;
      if (k < 3) {
        out[0] = out[1] = out[2] = NAN;
        continue;
      }
      float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      float K[3] = {0, 0, 0};
      if (centred) {
        K[0] = pts[3 * nb[0].idx];
        K[1] = pts[3 * nb[0].idx + 1];
        K[2] = pts[3 * nb[0].idx + 2];
      }
      for (int j = 0; j < k; ++j) {
        const float x = pts[3 * nb[j].idx] - K[0], y = pts[3 * nb[j].idx + 1] - K[1], z = pts[3 * nb[j].idx + 2] - K[2];
        if (fused) {
          acc[0] = fmaf(x, x, acc[0]); acc[1] = fmaf(x, y, acc[1]); acc[2] = fmaf(x, z, acc[2]);
          acc[3] = fmaf(y, y, acc[3]); acc[4] = fmaf(y, z, acc[4]); acc[5] = fmaf(z, z, acc[5]);
        } else {
          acc[0] += x * x; acc[1] += x * y; acc[2] += x * z;
          acc[3] += y * y; acc[4] += y * z; acc[5] += z * z;
        }
        acc[6] += x; acc[7] += y; acc[8] += z;
      }
      for (int i = 0; i < 9; ++i) acc[i] /= (float)k;
      float cov[9];
      if (fused) {
        cov[0] = fmaf(-acc[6], acc[6], acc[0]);
        cov[1] = fmaf(-acc[6], acc[7], acc[1]);
        cov[2] = fmaf(-acc[6], acc[8], acc[2]);
        cov[4] = fmaf(-acc[7], acc[7], acc[3]);
        cov[5] = fmaf(-acc[7], acc[8], acc[4]);
        cov[8] = fmaf(-acc[8], acc[8], acc[5]);
      } else {
        cov[0] = acc[0] - acc[6] * acc[6];
        cov[1] = acc[1] - acc[6] * acc[7];
        cov[2] = acc[2] - acc[6] * acc[8];
        cov[4] = acc[3] - acc[7] * acc[7];
        cov[5] = acc[4] - acc[7] * acc[8];
        cov[8] = acc[5] - acc[8] * acc[8];
      }
      cov[3] = cov[1];
      cov[6] = cov[2];
      cov[7] = cov[5];
      float ev, nv[3];
      eigen33_smallest(cov, &ev, nv);
      /* flipNormalTowardsViewpoint(point, 0, 0, 0, ...): vp - point */
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
    free(nb);
  }
  return 0;
}

/* Eigen::Vector4f dot product with the 4th component zero, in the order Eigen's SSE reduction adds the
 * four lane products: (x0 y0 + x2 y2) + (x1 y1 + x3 y3).  The summation order matters: the FPFH bins
 * are decided by floor() of these values and the fixture pins them to 1e-4. */
static inline float dot4(const float* x, const float* y) {
  const float p0 = x[0] * y[0], p1 = x[1] * y[1], p2 = x[2] * y[2];
  return (p0 + p2) + (p1 + 0.0f);
}

/* ---- decisions at the edge of float precision (pinning aid) ----------------------------------------------
 * Two kinds of DISCRETE decisions in the FPFH pipeline hang on the last bits of float expressions that PCL
 * evaluates with libm (acosf / atan2f / cosf / sinf; not portable to the last bit, and not what this
 * restatement's deterministic functions return):
 *   kind 0     "switch p1 and p2": acosf(|angle1|) > acosf(|angle2|) when the two cosines are a few ulps apart;
 *   kind 1..3  the histogram bin floor(11 x) of feature f1 / f2 / f3 when 11 x lies within ~1e-5 of an integer.
 * They are the only places where the restatement and the reference's fixtures (generated on x86-64 glibc) can
 * differ by more than rounding noise.  To pin the restatement on the fixtures at their own tolerance the tests can
 * (a) LIST every such near-boundary evaluation (p, q, kind, default outcome) and (b) FORCE listed evaluations
 * to the other outcome; everything else is untouched.  Default: nothing listed, nothing forced. */
static int32_t g_tie_collect = 0, g_tie_count = 0, g_tie_cap = 0;
static double g_bin_window = 0.0;
static int32_t* g_tie_out = NULL;           /* (p, q, kind, default outcome) quadruples */
static const int32_t* g_flip = NULL;        /* (p, q, kind) evaluations forced to the other outcome */
static int32_t g_flip_n = 0;
static _Thread_local int32_t t_pair_p = -1, t_pair_q = -1;
FEAT_API void feat_tie_hooks(int32_t switch_window_ulps, double bin_window, int32_t* out_quads, int32_t cap,
                             const int32_t* flip_triples, int32_t n_flip) {
  g_tie_collect = switch_window_ulps;
  g_bin_window = bin_window;
  g_tie_out = out_quads;
  g_tie_cap = cap;
  g_tie_count = 0;
  g_flip = flip_triples;
  g_flip_n = n_flip;
}
FEAT_API int32_t feat_tie_count(void) { return g_tie_count; }
static void tie_record(int32_t kind, int32_t outcome) {
  int32_t k;
#pragma omp atomic capture
  k = g_tie_count++;
  if (k < g_tie_cap) {
    g_tie_out[4 * k] = t_pair_p;
    g_tie_out[4 * k + 1] = t_pair_q;
    g_tie_out[4 * k + 2] = kind;
    g_tie_out[4 * k + 3] = outcome;
  }
}
static int tie_forced(int32_t kind) {
  for (int32_t k = 0; k < g_flip_n; ++k)
    if (g_flip[3 * k] == t_pair_p && g_flip[3 * k + 1] == t_pair_q && g_flip[3 * k + 2] == kind) return 1;
  return 0;
}

/* pcl::computePairFeatures (features/pfh_tools.hpp / pfh.hpp), float, Eigen::Vector4f arithmetic */
static int pair_features(const float* p1, const float* n1, const float* p2, const float* n2, float* f) {
  float dp[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  const float f4 = sqrtf(dot4(dp, dp));
  if (f4 == 0.0f) return 0;
  float a[3] = {n1[0], n1[1], n1[2]}, b[3] = {n2[0], n2[1], n2[2]};
  const float angle1 = dot4(a, dp) / f4;
  const float angle2 = dot4(b, dp) / f4;
  float f3;
  int sw = det_acosf(fabsf(angle1)) > det_acosf(fabsf(angle2));
  if (g_tie_collect || g_flip_n) {
    const float c1 = fabsf(angle1), c2 = fabsf(angle2);
    /* near-tie: the cosines are within a few ulps of each other (acos is monotone: far pairs cannot tie) */
    if (g_tie_collect && fabsf(c1 - c2) <= (float)g_tie_collect * 1.1920929e-07f * fmaxf(c1, c2)) tie_record(0, sw);
    if (g_flip_n && tie_forced(0)) sw = !sw;
  }
  if (sw) { /* switch p1 and p2 */
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
  cross3(dp, a, v);
  const float vn = sqrtf(dot4(v, v));
  if (vn == 0.0f) return 0;
  v[0] /= vn;
  v[1] /= vn;
  v[2] /= vn;
  float w[3];
  cross3(a, v, w);
  f[1] = dot4(v, b);
  f[0] = det_atan2f(dot4(w, b), dot4(a, b));
  f[2] = f3;
  f[3] = f4;
  return 1;
}

/* pcl::FPFHEstimation::computeFeature with setRadiusSearch(radius): SPFH of every point (11 + 11 + 11
 * bins), then the distance-weighted sum over the neighbours.  out: n x 33 floats. */
/* static_cast<int>(std::floor(x)) clamped to [0, 10]; a NaN feature (NaN normal) gives INT_MIN on x86,
 * i.e. bin 0 after the clamp -- stated explicitly so that every platform agrees */
static int fpfh_bin(double x) {
  if (!(x == x)) return 0;
  const double fl = floor(x);
  return fl < 0.0 ? 0 : (fl >= 11.0 ? 10 : (int)fl);
}
/* the same with the pinning hooks: `kind` = 1 + feature index */
static int fpfh_bin_hooked(double x, int kind) {
  int b = fpfh_bin(x);
  if ((g_bin_window > 0.0 || g_flip_n) && x == x) {
    const double r = rint(x);
    if (g_bin_window > 0.0 && fabs(x - r) <= g_bin_window && r >= 1.0 && r <= 10.0) tie_record(kind, b);
    if (g_flip_n && r >= 1.0 && r <= 10.0 && tie_forced(kind)) b = (x >= r) ? (int)r - 1 : (int)r; /* across the edge */
  }
  return b;
}

FEAT_API int feat_compute_fpfh(const float* pts, const float* normals, int32_t n, double radius, float* out) {
  float* spfh = (float*)calloc((size_t)n * 33, sizeof(float));
  if (!spfh) return 2;
  const float d_pi = 1.0f / (2.0f * (float)M_PI);
#pragma omp parallel
  {
    nbr_t* nb = (nbr_t*)malloc((size_t)n * sizeof(nbr_t));
#pragma omp for schedule(dynamic, 16)
    for (int p = 0; p < n; ++p) { /* computePointSPFHSignature */
      const int k = radius_search(pts, n, p, radius, nb);
      if (k == 0) continue;
      const float incr = 100.0f / (float)(k - 1);
      float* h = spfh + (size_t)p * 33;
      for (int j = 0; j < k; ++j) {
        const int qi = nb[j].idx;
        if (qi == p) continue;
        float f[4];
        t_pair_p = p;
        t_pair_q = qi;
        if (!pair_features(pts + 3 * p, normals + 3 * p, pts + 3 * qi, normals + 3 * qi, f)) continue;
        h[fpfh_bin_hooked(11 * (((double)f[0] + M_PI) * (double)d_pi), 1)] += incr;
        h[11 + fpfh_bin_hooked(11 * (((double)f[1] + 1.0) * 0.5), 2)] += incr;
        h[22 + fpfh_bin_hooked(11 * (((double)f[2] + 1.0) * 0.5), 3)] += incr;
      }
    }
#pragma omp for schedule(dynamic, 16)
    for (int p = 0; p < n; ++p) { /* weightPointSPFHSignature */
      const int k = radius_search(pts, n, p, radius, nb);
      float* o = out + (size_t)p * 33;
      for (int i = 0; i < 33; ++i) o[i] = 0.0f;
      float sum[3] = {0, 0, 0};
      for (int j = 0; j < k; ++j) {
        if (nb[j].d2 == 0.0f) continue;
        const float weight = 1.0f / nb[j].d2;
        const float* h = spfh + (size_t)nb[j].idx * 33;
        for (int g = 0; g < 3; ++g)
          for (int i = 0; i < 11; ++i) {
            const float val = h[11 * g + i] * weight;
            sum[g] += val;
            o[11 * g + i] += val;
          }
      }
      for (int g = 0; g < 3; ++g) {
        if (sum[g] != 0) sum[g] = 100.0f / sum[g];
        for (int i = 0; i < 11; ++i) o[11 * g + i] *= sum[g];
      }
    }
    free(nb);
  }
  free(spfh);
  return 0;
}

/* exact L2 1-NN of every row of `query` (nq x dim) in `data` (nd x dim), float accumulation in index
 * order as flann::L2<float> does (4-way unrolled there; differences are at the 1e-7 level), ties to the
 * lowest index. */
FEAT_API int feat_nn1(const float* data, int32_t nd, const float* query, int32_t nq, int32_t dim, int32_t* nn) {
#pragma omp parallel for schedule(static)
  for (int q = 0; q < nq; ++q) {
    float best = INFINITY;
    int bi = -1;
    for (int i = 0; i < nd; ++i) {
      float d = 0;
      for (int c = 0; c < dim; ++c) {
        const float t = query[(size_t)q * dim + c] - data[(size_t)i * dim + c];
        d += t * t;
      }
      if (d < best) {
        best = d;
        bi = i;
      }
    }
    nn[q] = bi;
  }
  return 0;
}

static int cmp_pair(const void* a, const void* b) {
  const int32_t* x = (const int32_t*)a;
  const int32_t* y = (const int32_t*)b;
  if (x[0] != y[0]) return (x[0] > y[0]) - (x[0] < y[0]);
  return (x[1] > y[1]) - (x[1] < y[1]);
}

/* Matcher::calculateCorrespondences (matcher.cc:21-301) for use_tuple_test = false (the tuple test draws
 * from rand() seeded with time(NULL), matcher.cc:214: not reproducible by construction; every reference
 * caller passes false).  out: room for 2 * (n_src + n_dst) int32 pairs; returns the pair count. */
FEAT_API int32_t feat_match(const float* src_feat, int32_t n_src, const float* dst_feat, int32_t n_dst,
                            int32_t dim, int32_t use_crosscheck, int32_t* out) {
  /* advancedMatching: i = the larger cloud, j = the smaller (matcher.cc:123-133) */
  int swapped = n_dst > n_src;
  const float* fi = swapped ? dst_feat : src_feat;
  const float* fj = swapped ? src_feat : dst_feat;
  const int ni = swapped ? n_dst : n_src, nj = swapped ? n_src : n_dst;
  int32_t* j_to_i = (int32_t*)malloc((size_t)nj * sizeof(int32_t));
  int32_t* i_nn = (int32_t*)malloc((size_t)ni * sizeof(int32_t));
  int32_t* i_to_j = (int32_t*)malloc((size_t)ni * sizeof(int32_t));
  feat_nn1(fi, ni, fj, nj, dim, j_to_i); /* :162 for every j its nearest i */
  feat_nn1(fj, nj, fi, ni, dim, i_nn);   /* :165 (evaluated lazily there, same values) */
  for (int i = 0; i < ni; ++i) i_to_j[i] = -1;
  for (int j = 0; j < nj; ++j) {
    const int i = j_to_i[j];
    if (i_to_j[i] == -1) i_to_j[i] = i_nn[i];
  }
  int32_t cnt = 0;
  if (use_crosscheck) {
    /* :198-233: (i, j) with i -> j in corres_ij and j -> i in corres_ji */
    for (int i = 0; i < ni; ++i) {
      const int j = i_to_j[i];
      if (j >= 0 && j_to_i[j] == i) {
        out[2 * cnt] = i;
        out[2 * cnt + 1] = j;
        ++cnt;
      }
    }
  } else {
    for (int i = 0; i < ni; ++i)
      if (i_to_j[i] != -1) {
        out[2 * cnt] = i;
        out[2 * cnt + 1] = i_to_j[i];
        ++cnt;
      }
    for (int j = 0; j < nj; ++j) {
      out[2 * cnt] = j_to_i[j];
      out[2 * cnt + 1] = j;
      ++cnt;
    }
  }
  if (swapped) /* :281-287 */
    for (int32_t k = 0; k < cnt; ++k) {
      const int32_t t = out[2 * k];
      out[2 * k] = out[2 * k + 1];
      out[2 * k + 1] = t;
    }
  qsort(out, (size_t)cnt, 2 * sizeof(int32_t), cmp_pair); /* :295-296 sort + unique */
  int32_t u = 0;
  for (int32_t k = 0; k < cnt; ++k)
    if (u == 0 || out[2 * k] != out[2 * (u - 1)] || out[2 * k + 1] != out[2 * (u - 1) + 1]) {
      out[2 * u] = out[2 * k];
      out[2 * u + 1] = out[2 * k + 1];
      ++u;
    }
  free(j_to_i);
  free(i_nn);
  free(i_to_j);
  return u;
}
