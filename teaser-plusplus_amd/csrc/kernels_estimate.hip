// kernels_estimate.hip -- K5 (GNC-TLS rotation) and K6 (scalar TLS / translation) for gfx950.
//
// These stages run on K = |max clique| columns only (reference registration.cc:657-731): they
// are latency-bound, so each is ONE launch with one workgroup per problem and the whole GNC loop,
// including the 3x3 SVD, inside the kernel (no host round trips).
//   K5: GNCTLSRotationSolver::solveForRotation, reference registration.cc:764-866, with
//       utils::svdRot, reference utils.h:121-136.
//   K6: ScalarTLSEstimator::estimate, reference registration.cc:21-88, and
//       TLSTranslationSolver::solveForTranslation, reference registration.cc:445-471.
#include <math.h>

#include <type_traits>

#include "internal.h"

namespace thip {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    double t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}

// ------------------------------------------------------------------------------------------
// 3x3 SVD rotation (utils.h:121-136): R = V diag(1,1,det(U)det(V)) U^T, H = U S V^T.
// One-sided (Hestenes) Jacobi on the columns of H; row-major 3x3 arrays; run by one thread.
// ------------------------------------------------------------------------------------------
__device__ double det3(const double* M) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
         M[2] * (M[3] * M[7] - M[4] * M[6]);
}

__device__ void svd_rot3(const double* H, double* R) {
  double B[9], V[9];
  for (int i = 0; i < 9; ++i) {
    B[i] = H[i];
    V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  }
  for (int sweep = 0; sweep < 60; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < 2; ++p) {
      for (int q = p + 1; q < 3; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < 3; ++r) {
          alpha += B[3 * r + p] * B[3 * r + p];
          beta += B[3 * r + q] * B[3 * r + q];
          gamma += B[3 * r + p] * B[3 * r + q];
        }
        // converged pair: the columns are orthogonal to working precision.  (The threshold used to be 1e-17, below
        // the rounding noise of gamma itself (~1e-16 sqrt(alpha beta)): most calls then ran all 60 sweeps -- ~50 us
        // of one thread's FP64 divisions and square roots per GNC iteration, the larger part of the rotation stage.)
        if (gamma == 0.0 || fabs(gamma) <= 1e-15 * sqrt(alpha * beta)) continue;
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int r = 0; r < 3; ++r) {
          const double bp = B[3 * r + p], bq = B[3 * r + q];
          B[3 * r + p] = c * bp - s * bq;
          B[3 * r + q] = s * bp + c * bq;
          const double vp = V[3 * r + p], vq = V[3 * r + q];
          V[3 * r + p] = c * vp - s * vq;
          V[3 * r + q] = s * vp + c * vq;
        }
      }
    }
    if (!rotated) break;
  }
  double sig[3];
  for (int c = 0; c < 3; ++c)
    sig[c] = sqrt(B[c] * B[c] + B[3 + c] * B[3 + c] + B[6 + c] * B[6 + c]);
  // order columns by descending singular value
  int i0 = 0, i1 = 1, i2 = 2;
  if (sig[i1] > sig[i0]) { int t = i0; i0 = i1; i1 = t; }
  if (sig[i2] > sig[i0]) { int t = i0; i0 = i2; i2 = t; }
  if (sig[i2] > sig[i1]) { int t = i1; i1 = i2; i2 = t; }
  const double s0 = sig[i0], s1 = sig[i1], s2 = sig[i2];
  double u0[3], u1[3], u2[3], v0[3], v1[3], v2[3];
  for (int r = 0; r < 3; ++r) {
    v0[r] = V[3 * r + i0];
    v1[r] = V[3 * r + i1];
    v2[r] = V[3 * r + i2];
  }
  const double tiny = 1e-300;
  if (s0 > tiny) {
    for (int r = 0; r < 3; ++r) u0[r] = B[3 * r + i0] / s0;
  } else {
    u0[0] = 1; u0[1] = 0; u0[2] = 0;
  }
  if (s1 > tiny && s1 > 1e-15 * s0) {
    for (int r = 0; r < 3; ++r) u1[r] = B[3 * r + i1] / s1;
  } else {
    const int k = (fabs(u0[0]) <= fabs(u0[1]) && fabs(u0[0]) <= fabs(u0[2])) ? 0
                  : (fabs(u0[1]) <= fabs(u0[2]) ? 1 : 2);
    double nn = 0;
    for (int r = 0; r < 3; ++r) {
      u1[r] = (r == k ? 1.0 : 0.0) - u0[k] * u0[r];
      nn += u1[r] * u1[r];
    }
    nn = sqrt(nn);
    for (int r = 0; r < 3; ++r) u1[r] /= nn;
  }
  if (s2 > tiny && s2 > 1e-15 * s0) {
    for (int r = 0; r < 3; ++r) u2[r] = B[3 * r + i2] / s2;
  } else {
    u2[0] = u0[1] * u1[2] - u0[2] * u1[1];
    u2[1] = u0[2] * u1[0] - u0[0] * u1[2];
    u2[2] = u0[0] * u1[1] - u0[1] * u1[0];
  }
  double U[9], Vs[9];
  for (int r = 0; r < 3; ++r) {
    U[3 * r] = u0[r]; U[3 * r + 1] = u1[r]; U[3 * r + 2] = u2[r];
    Vs[3 * r] = v0[r]; Vs[3 * r + 1] = v1[r]; Vs[3 * r + 2] = v2[r];
  }
  if (det3(U) * det3(Vs) < 0) {  // utils.h:131-133
    Vs[2] = -Vs[2]; Vs[5] = -Vs[5]; Vs[8] = -Vs[8];
  }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      R[3 * r + c] = Vs[3 * r] * U[3 * c] + Vs[3 * r + 1] * U[3 * c + 1] + Vs[3 * r + 2] * U[3 * c + 2];
}

// ------------------------------------------------------------------------------------------
// TIMs handed to the rotation solver (registration.cc:657-697), generated on the fly.
// mode 0 CHAIN over the sorted clique, 1 COMPLETE (computeTIMs pair order), 2 raw columns.
// ------------------------------------------------------------------------------------------
struct TimSource {
  const double* ps;
  const double* pd;
  const int32_t* c;
  int K;
  int mode;
  double inv_scale;
  __device__ __forceinline__ void get(int64_t j, double* x, double* y) const {
    if (mode == 2) {
      for (int r = 0; r < 3; ++r) {
        x[r] = ps[3 * j + r];
        y[r] = pd[3 * j + r];
      }
      return;
    }
    int a, b;  // TIM = v_b - v_a
    if (mode == 0) {
      a = (int)j;
      b = (j + 1 == K) ? 0 : (int)j + 1;  // registration.cc:665-671: leaf - root
    } else {
      // invert k = a*K - a(a+1)/2 + (b-a-1), registration.cc:531
      const double kk = 2.0 * K - 1.0;
      int aa = (int)floor((kk - sqrt(kk * kk - 8.0 * (double)j)) * 0.5);
      if (aa < 0) aa = 0;
      while ((int64_t)(aa + 1) * K - (int64_t)(aa + 1) * (aa + 2) / 2 <= j) ++aa;
      while ((int64_t)aa * K - (int64_t)aa * (aa + 1) / 2 > j) --aa;
      a = aa;
      b = (int)(j - ((int64_t)aa * K - (int64_t)aa * (aa + 1) / 2)) + aa + 1;
    }
    const int64_t ia = c[a], ib = c[b];
    for (int r = 0; r < 3; ++r) {
      x[r] = ps[3 * ib + r] - ps[3 * ia + r];
      y[r] = (pd[3 * ib + r] - pd[3 * ia + r]) * inv_scale;  // registration.cc:697
    }
  }
};

__device__ __forceinline__ double residual_sq(const double* R, const double* x, const double* y) {
  double s = 0;  // registration.cc:812-813
  for (int r = 0; r < 3; ++r) {
    const double d = y[r] - (R[3 * r] * x[0] + R[3 * r + 1] * x[1] + R[3 * r + 2] * x[2]);
    s += d * d;
  }
  return s;
}

// The GNC-TLS loop for one problem, executed by a 256-thread workgroup.
__device__ void gnc_tls_block(const TimSource& ts, int64_t KT, double noise_bound,
                              double gnc_factor, int64_t max_iterations, double cost_threshold,
                              double* w, double* R_out, double* cost_out, int* iters_out,
                              double* sh /* LDS: 64 doubles */) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* sH = sh;        // [4][9] partials -> H in sH[0..8]
  double* sR = sh + 36;   // 9
  double* sS = sh + 45;   // 4 partials + scalars
  double noise_bound_sq = noise_bound * noise_bound;  // registration.cc:793-796
  if (noise_bound_sq < 1e-16) noise_bound_sq = 1e-2;
  for (int64_t j = tid; j < KT; j += 256) w[j] = 1.0;
  double mu = 1, prev_cost = INFINITY, cost = INFINITY;
  int iters = 0;
  __syncthreads();
  for (int64_t it = 0; it < max_iterations; ++it) {
    ++iters;
    // H = X diag(w) Y^T   (utils.h:125)
    double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t j = tid; j < KT; j += 256) {
      double x[3], y[3];
      ts.get(j, x, y);
      const double wj = w[j];
      for (int r = 0; r < 3; ++r) {
        const double xw = x[r] * wj;
        h[3 * r] += xw * y[0];
        h[3 * r + 1] += xw * y[1];
        h[3 * r + 2] += xw * y[2];
      }
    }
    for (int k = 0; k < 9; ++k) {
      const double v = wave_sum_d(h[k]);
      if (lane == 0) sH[wave * 9 + k] = v;
    }
    __syncthreads();
    if (tid == 0) {
      double H[9];
      for (int k = 0; k < 9; ++k) H[k] = (sH[k] + sH[9 + k]) + (sH[18 + k] + sH[27 + k]);
      svd_rot3(H, sR);  // registration.cc:809
    }
    __syncthreads();
    double R[9];
    for (int k = 0; k < 9; ++k) R[k] = sR[k];
    if (it == 0) {  // registration.cc:814-825
      double mx = -INFINITY;
      for (int64_t j = tid; j < KT; j += 256) {
        double x[3], y[3];
        ts.get(j, x, y);
        const double r2 = residual_sq(R, x, y);
        mx = r2 > mx ? r2 : mx;
      }
      mx = wave_max_d(mx);
      if (lane == 0) sS[wave] = mx;
      __syncthreads();
      const double m01 = sS[0] > sS[1] ? sS[0] : sS[1];
      const double m23 = sS[2] > sS[3] ? sS[2] : sS[3];
      const double max_residual = m01 > m23 ? m01 : m23;
      mu = 1 / (2 * max_residual / noise_bound_sq - 1);
      __syncthreads();
      if (mu <= 0) break;
    }
    const double th1 = (mu + 1) / mu * noise_bound_sq;  // registration.cc:828-829
    const double th2 = mu / (mu + 1) * noise_bound_sq;
    double c = 0;
    for (int64_t j = tid; j < KT; j += 256) {  // registration.cc:831-844
      double x[3], y[3];
      ts.get(j, x, y);
      const double r2 = residual_sq(R, x, y);
      c += w[j] * r2;
      double nw;
      if (r2 >= th1) {
        nw = 0;
      } else if (r2 <= th2) {
        nw = 1;
      } else {
        nw = sqrt(noise_bound_sq * mu * (mu + 1) / r2) - mu;
      }
      w[j] = nw;
    }
    c = wave_sum_d(c);
    if (lane == 0) sS[wave] = c;
    __syncthreads();
    cost = (sS[0] + sS[1]) + (sS[2] + sS[3]);
    __syncthreads();
    const double cost_diff = fabs(cost - prev_cost);  // registration.cc:847-858
    mu = mu * gnc_factor;
    prev_cost = cost;
    if (cost_diff < cost_threshold) break;
  }
  if (tid == 0) {
    for (int k = 0; k < 9; ++k) R_out[k] = sR[k];
    *cost_out = cost;
    *iters_out = iters;
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------
// FastGlobalRegistrationSolver::solveForRotation (reference registration.cc:206-278) for one
// problem, executed by a 256-thread workgroup; same interface as gnc_tls_block.  w[] holds the
// line-process weights l_pq; the caller derives the inlier mask l_pq != 0 (:277-279).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double block256_sum(double v, double* s4 /* LDS 4 */) {
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
  __syncthreads();
  return (s4[0] + s4[1]) + (s4[2] + s4[3]);
}
__device__ __forceinline__ double block256_max(double v, double* s4) {
  v = wave_max_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
  __syncthreads();
  const double a = s4[0] > s4[1] ? s4[0] : s4[1], b = s4[2] > s4[3] ? s4[2] : s4[3];
  return a > b ? a : b;
}

__device__ void fgr_block(const TimSource& ts, int64_t KT, double noise_bound, double gnc_factor,
                          int64_t max_iterations, double cost_threshold, double* w, double* R_out,
                          double* cost_out, int* iters_out, double* sh /* LDS: 64 doubles */) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* sH = sh;        // [4][9]
  double* sR = sh + 36;   // 9
  double* sS = sh + 45;   // 4
  const double noise_bound_sq = noise_bound * noise_bound;  // :221
  // utils::calculateDiameter (utils.h:107-112) of both TIM sets; the reference returns FLOAT
  double cg[6] = {0, 0, 0, 0, 0, 0};
  for (int64_t j = tid; j < KT; j += 256) {
    double x[3], y[3];
    ts.get(j, x, y);
    for (int r = 0; r < 3; ++r) {
      cg[r] += x[r];
      cg[3 + r] += y[r];
    }
  }
  for (int k = 0; k < 6; ++k) cg[k] = block256_sum(cg[k], sS) / (double)KT;
  double mxs = -INFINITY, mxd = -INFINITY;
  for (int64_t j = tid; j < KT; j += 256) {
    double x[3], y[3];
    ts.get(j, x, y);
    double a = 0, b = 0;
    for (int r = 0; r < 3; ++r) {
      a += (x[r] - cg[r]) * (x[r] - cg[r]);
      b += (y[r] - cg[3 + r]) * (y[r] - cg[3 + r]);
    }
    mxs = a > mxs ? a : mxs;
    mxd = b > mxd ? b : mxd;
  }
  mxs = block256_max(mxs, sS);
  mxd = block256_max(mxd, sS);
  const double src_diameter = (double)(float)(2 * sqrt(mxs));  // :226-227
  const double dest_diameter = (double)(float)(2 * sqrt(mxd));
  double global_scale = src_diameter > dest_diameter ? src_diameter : dest_diameter;  // :228
  global_scale /= noise_bound_sq;                                                      // :229
  double mu = global_scale * global_scale / noise_bound_sq;                            // :230
  const double min_mu = 1.0;                                                           // :233
  if (tid < 9) sR[tid] = (tid % 4 == 0) ? 1.0 : 0.0;                                   // :234
  for (int64_t j = tid; j < KT; j += 256) w[j] = 1.0;                                  // :235-236
  double cost = INFINITY;                                                              // :223
  int iters = 0;
  __syncthreads();
  for (int64_t it = 0; it < max_iterations; ++it) {  // :242
    ++iters;
    const double scaled_mu = mu * noise_bound_sq;  // :243
    double R[9];
    for (int k = 0; k < 9; ++k) R[k] = sR[k];
    __syncthreads();
    // :247-253 line-process weights from the current R, fused with H = X diag(l) Y^T (:256)
    double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t j = tid; j < KT; j += 256) {
      double x[3], y[3];
      ts.get(j, x, y);
      const double q = scaled_mu / (scaled_mu + residual_sq(R, x, y));
      const double l = q * q;
      w[j] = l;
      for (int r = 0; r < 3; ++r) {
        const double xw = x[r] * l;
        h[3 * r] += xw * y[0];
        h[3 * r + 1] += xw * y[1];
        h[3 * r + 2] += xw * y[2];
      }
    }
    for (int k = 0; k < 9; ++k) {
      const double v = wave_sum_d(h[k]);
      if (lane == 0) sH[wave * 9 + k] = v;
    }
    __syncthreads();
    if (tid == 0) {
      double H[9];
      for (int k = 0; k < 9; ++k) H[k] = (sH[k] + sH[9 + k]) + (sH[18 + k] + sH[27 + k]);
      svd_rot3(H, sR);
    }
    __syncthreads();
    for (int k = 0; k < 9; ++k) R[k] = sR[k];
    double c = 0;  // :259-262
    for (int64_t j = tid; j < KT; j += 256) {
      double x[3], y[3];
      ts.get(j, x, y);
      const double d = residual_sq(R, x, y);
      c += (scaled_mu * d) / (scaled_mu + d);
    }
    cost = block256_sum(c, sS);
    if (cost < cost_threshold || mu < min_mu) break;  // :265-271
    mu /= gnc_factor;                                  // :274
  }
  __syncthreads();
  if (tid == 0) {
    for (int k = 0; k < 9; ++k) R_out[k] = sR[k];
    *cost_out = cost;
    *iters_out = iters;
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------
// QuatroSolver::solveForRotation (reference registration.cc:280-408): GNC-TLS on the xy rows only
// (yaw), utils::svdRot2d (utils.h:145-160).  The 2x2 polar rotation V diag(1,det) U^T of
// H = X diag(w) Y^T is the rotation by atan2(H01 - H10, H00 + H11).  The reference keeps the noise
// bound in function-local statics (:329-330, first call in the process wins); here the per-solve
// value is used, which is what a fresh process computes.
// ------------------------------------------------------------------------------------------
__device__ void quatro_block(const TimSource& ts, int64_t KT, double noise_bound,
                             double gnc_factor, int64_t max_iterations, double cost_threshold,
                             double* w, double* R_out, double* cost_out, int* iters_out,
                             double* sh /* LDS: 64 doubles */) {
  const int tid = threadIdx.x;
  double* sS = sh + 45;
  double noise_bound_sq = noise_bound * noise_bound;
  if (noise_bound_sq < 1e-16) noise_bound_sq = 1e-2;  // :331-333
  for (int64_t j = tid; j < KT; j += 256) w[j] = 1.0;
  double mu = 1, prev_cost = INFINITY, cost = INFINITY;  // :324-327
  double c2 = 1, s2 = 0;                                 // rotation_2d = identity (:303)
  int iters = 0;
  __syncthreads();
  for (int64_t it = 0; it < max_iterations; ++it) {  // :343
    ++iters;
    double h[4] = {0, 0, 0, 0};  // :346
    for (int64_t j = tid; j < KT; j += 256) {
      double x[3], y[3];
      ts.get(j, x, y);
      const double wj = w[j];
      h[0] += x[0] * wj * y[0];
      h[1] += x[0] * wj * y[1];
      h[2] += x[1] * wj * y[0];
      h[3] += x[1] * wj * y[1];
    }
    for (int k = 0; k < 4; ++k) h[k] = block256_sum(h[k], sS);
    {
      const double a = h[0] + h[3], b = h[1] - h[2], nrm = sqrt(a * a + b * b);
      c2 = 1;
      s2 = 0;
      if (nrm > 0) {
        c2 = a / nrm;
        s2 = b / nrm;
      }
    }
    if (it == 0) {  // :351-362
      double mx = -INFINITY;
      for (int64_t j = tid; j < KT; j += 256) {
        double x[3], y[3];
        ts.get(j, x, y);
        const double d0 = y[0] - (c2 * x[0] - s2 * x[1]), d1 = y[1] - (s2 * x[0] + c2 * x[1]);
        const double r2 = d0 * d0 + d1 * d1;
        mx = r2 > mx ? r2 : mx;
      }
      const double max_residual = block256_max(mx, sS);
      mu = 1 / (2 * max_residual / noise_bound_sq - 1);
      if (mu <= 0) break;
    }
    const double th1 = (mu + 1) / mu * noise_bound_sq;  // :365-366
    const double th2 = mu / (mu + 1) * noise_bound_sq;
    double c = 0;
    for (int64_t j = tid; j < KT; j += 256) {  // :368-381
      double x[3], y[3];
      ts.get(j, x, y);
      const double d0 = y[0] - (c2 * x[0] - s2 * x[1]), d1 = y[1] - (s2 * x[0] + c2 * x[1]);
      const double r2 = d0 * d0 + d1 * d1;
      c += w[j] * r2;
      double nw;
      if (r2 >= th1) {
        nw = 0;
      } else if (r2 <= th2) {
        nw = 1;
      } else {
        nw = sqrt(noise_bound_sq * mu * (mu + 1) / r2) - mu;
      }
      w[j] = nw;
    }
    cost = block256_sum(c, sS);
    const double cost_diff = fabs(cost - prev_cost);  // :384-395
    mu = mu * gnc_factor;
    prev_cost = cost;
    if (cost_diff < cost_threshold) break;
  }
  __syncthreads();
  if (tid == 0) {  // :292 + :407
    R_out[0] = c2; R_out[1] = -s2; R_out[2] = 0;
    R_out[3] = s2; R_out[4] = c2;  R_out[5] = 0;
    R_out[6] = 0;  R_out[7] = 0;   R_out[8] = 1;
    *cost_out = cost;
    *iters_out = iters;
  }
  __syncthreads();
}

// rotation_estimation_algorithm dispatch (registration.h:856-869); returns the weight threshold
// that defines the rotation inlier mask: GNC-TLS w >= 0.5 (:861-865), FGR l_pq != 0 (:277-279,
// expressed as w > 0), QUATRO w >= 0.4 (:398-402).
__device__ void rotation_block(int algorithm, const TimSource& ts, int64_t KT, double noise_bound,
                               const EstParams& ep, double* w, double* R_out, double* cost_out,
                               int* iters_out, double* sh) {
  if (algorithm == TEASER_ROT_FGR)
    fgr_block(ts, KT, noise_bound, ep.gnc_factor, ep.max_iterations, ep.cost_threshold, w, R_out,
              cost_out, iters_out, sh);
  else if (algorithm == TEASER_ROT_QUATRO)
    quatro_block(ts, KT, noise_bound, ep.gnc_factor, ep.max_iterations, ep.cost_threshold, w, R_out,
                 cost_out, iters_out, sh);
  else
    gnc_tls_block(ts, KT, noise_bound, ep.gnc_factor, ep.max_iterations, ep.cost_threshold, w,
                  R_out, cost_out, iters_out, sh);
}
__device__ __forceinline__ bool rotation_inlier(int algorithm, double wj) {
  if (algorithm == TEASER_ROT_FGR) return wj != 0.0;
  if (algorithm == TEASER_ROT_QUATRO) return wj >= 0.4;
  return wj >= 0.5;
}

// Ordered compaction of { j : pred(j) } into out[] by the first 256 threads of the workgroup
// (every thread of the block must call: the barriers are block-wide); returns the count.
template <typename Pred>
__device__ int block_compact(int64_t n, Pred pred, int32_t* out, int* scan /* LDS 257 ints */) {
  const int tid = threadIdx.x;
  const bool active = tid < 256;
  const int64_t L = (n + 255) / 256;
  const int64_t b = active ? tid * L : n, e = (b + L < n) ? b + L : n;
  int cnt = 0;
  for (int64_t j = b; j < e; ++j) cnt += pred(j) ? 1 : 0;
  if (active) scan[tid] = cnt;
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int t = 0; t < 256; ++t) {
      const int v = scan[t];
      scan[t] = acc;
      acc += v;
    }
    scan[256] = acc;
  }
  __syncthreads();
  if (active && out) {
    int pos = scan[tid];
    for (int64_t j = b; j < e; ++j)
      if (pred(j)) out[pos++] = (int32_t)j;
  }
  const int total = scan[256];
  __syncthreads();
  return total;
}

__global__ __launch_bounds__(256) void gnc_tls_kernel(const ProbDesc* __restrict__ descs,
                                                      const double* __restrict__ src,
                                                      const double* __restrict__ dst,
                                                      const int32_t* __restrict__ clique,
                                                      ProbState* __restrict__ states, EstParams ep,
                                                      double* __restrict__ weights,
                                                      int32_t* __restrict__ rot_inliers,
                                                      const int64_t* __restrict__ tim_off) {
  TAIL_WAVE_PRIO();
  __shared__ double sh[64];
  __shared__ int scan[257];
  const ProbDesc d = descs[blockIdx.x];
  ProbState* st = states + blockIdx.x;
  const int K = st->clique_size;
  if (K <= 1) {  // registration.cc:643-647: invalid, rotation untouched
    if (threadIdx.x == 0) {
      st->n_rot = 0;
      st->gnc_iters = 0;
    }
    return;
  }
  TimSource ts;
  ts.ps = src + 3 * d.pt_off;
  ts.pd = dst + 3 * d.pt_off;
  ts.c = clique + d.pt_off;
  ts.K = K;
  ts.mode = ep.tim_graph;
  const double scale = st->scale;
  ts.inv_scale = 1 / scale;
  const int64_t KT = ep.tim_graph == 0 ? (int64_t)K : (int64_t)K * (K - 1) / 2;
  double* w = weights + tim_off[blockIdx.x];
  const double nb = ep.noise_bound * (2 / scale);  // registration.cc:702-704
  rotation_block(ep.algorithm, ts, KT, nb, ep, w, st->R, &st->gnc_cost, &st->gnc_iters, sh);
  // registration.cc:861-865 (and the FGR / QUATRO masks) + :712-716
  const int alg = ep.algorithm;
  const int cnt = block_compact(KT, [&](int64_t j) { return rotation_inlier(alg, w[j]); },
                                rot_inliers + tim_off[blockIdx.x], scan);
  if (threadIdx.x == 0) st->n_rot = cnt;
}

void launch_gnc_tls(hipStream_t s, const ProbDesc* d_desc, int batch, const double* d_src,
                    const double* d_dst, const int32_t* d_clique, ProbState* d_state,
                    EstParams ep, double* d_weights, int32_t* d_rot_inliers,
                    const int64_t* d_tim_off) {
  if (batch <= 0) return;
  hipLaunchKernelGGL(gnc_tls_kernel, dim3(batch), dim3(256), 0, s, d_desc, d_src, d_dst, d_clique,
                     d_state, ep, d_weights, d_rot_inliers, d_tim_off);
}

// Stand-alone rotation stage (solveForRotation entry point): src/dst are raw 3xK TIMs on device.
__global__ __launch_bounds__(256) void gnc_tls_raw_kernel(const double* __restrict__ src,
                                                          const double* __restrict__ dst, int K,
                                                          double noise_bound, EstParams ep,
                                                          double* __restrict__ w,
                                                          double* __restrict__ out /* R9,cost */,
                                                          int32_t* __restrict__ iters) {
  __shared__ double sh[64];
  TimSource ts;
  ts.ps = src;
  ts.pd = dst;
  ts.c = nullptr;
  ts.K = K;
  ts.mode = 2;
  ts.inv_scale = 1;
  rotation_block(ep.algorithm, ts, K, noise_bound, ep, w, out, out + 9, iters, sh);
  // the caller turns w into the inlier mask: rewrite it as 1 / 0 so one threshold (>= 0.5) fits all
  __syncthreads();
  for (int j = threadIdx.x; j < K; j += 256) w[j] = rotation_inlier(ep.algorithm, w[j]) ? 1.0 : 0.0;
}

void launch_gnc_tls_raw(hipStream_t s, const double* d_src, const double* d_dst, int K,
                        double noise_bound, EstParams ep, double* d_w, double* d_out,
                        int32_t* d_iters) {
  hipLaunchKernelGGL(gnc_tls_raw_kernel, dim3(1), dim3(256), 0, s, d_src, d_dst, K, noise_bound,
                     ep, d_w, d_out, d_iters);
}

// ------------------------------------------------------------------------------------------
// Scalar TLS (registration.cc:21-88) by a group of 256 threads (`gt` = thread index in the
// group).  Endpoints (value, tag) with tag = +(i+1) for x_i - r_i, -(i+1) for x_i + r_i
// (registration.cc:35-38) are bitonic-sorted ascending by (value, insertion index) -- the
// reference's std::sort leaves tie order unspecified; insertion order is what a stable sort
// gives.  The sweep (registration.cc:58-75) becomes a blocked prefix sum: each thread owns a
// contiguous run of endpoints, run totals are combined sequentially in run order.
// All __syncthreads() are block-wide: every group of the block must call with the same n.
// ------------------------------------------------------------------------------------------
struct TlsScratch {
  double* val;   // [P2]
  int32_t* tag;  // [P2]
  double* tot;   // [6][256] run totals / prefixes
  double* best;  // [256] cost, [256] x_hat
  int32_t* bpos; // [256]
};

__device__ __forceinline__ bool ep_less(double va, int ta, double vb, int tb) {
  if (va < vb) return true;
  if (va > vb) return false;
  // insertion index: 2*i for the opening endpoint, 2*i+1 for the closing one
  const unsigned int oa = ta > 0 ? 2u * (unsigned)(ta - 1) : 2u * (unsigned)(-ta - 1) + 1u;
  const unsigned int ob = tb > 0 ? 2u * (unsigned)(tb - 1) : 2u * (unsigned)(-tb - 1) + 1u;
  return oa < ob;
}

__device__ double scalar_tls_group(const double* X, const double* Rg, double r_const, int n,
                                   const TlsScratch& sc, int gt) {
  const int m = 2 * n;
  int P2 = 2;
  while (P2 < m) P2 <<= 1;
  for (int e = gt; e < P2; e += 256) {
    if (e < m) {
      const int i = e >> 1;
      const double r = Rg ? Rg[i] : r_const;
      sc.val[e] = (e & 1) ? X[i] + r : X[i] - r;
      sc.tag[e] = (e & 1) ? -(i + 1) : (i + 1);
    } else {
      sc.val[e] = INFINITY;
      sc.tag[e] = 0x7fffffff;
    }
  }
  __syncthreads();
  for (int k = 2; k <= P2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = gt; i < P2; i += 256) {
        const int l = i ^ j;
        if (l > i) {
          const double vi = sc.val[i], vl = sc.val[l];
          const int ti = sc.tag[i], tl = sc.tag[l];
          const bool up = (i & k) == 0;
          const bool l_lt_i = ep_less(vl, tl, vi, ti);
          if (l_lt_i == up) {
            sc.val[i] = vl; sc.val[l] = vi;
            sc.tag[i] = tl; sc.tag[l] = ti;
          }
        }
      }
      __syncthreads();
    }
  }
  // run totals
  const int L = (m + 255) / 256;
  const int b = gt * L, e_end = min(m, b + L);
  double t_dwc = 0, t_dxw = 0, t_ris = 0, t_sx = 0, t_sxx = 0, t_card = 0;
  for (int e = b; e < e_end; ++e) {
    const int tg = sc.tag[e];
    const int idx = (tg > 0 ? tg : -tg) - 1;
    const double eps = tg > 0 ? 1.0 : -1.0;
    const double x = X[idx], r = Rg ? Rg[idx] : r_const;
    const double w = 1.0 / (r * r);
    t_card += eps;
    t_dwc += eps * w;
    t_dxw += eps * w * x;
    t_ris += eps * r;
    t_sx += eps * x;
    t_sxx += eps * x * x;
  }
  // sum of all ranges (registration.cc:51) as a run-ordered sum too
  double rsum = 0;
  {
    const int Ln = (n + 255) / 256;
    const int bn = gt * Ln, en = min(n, bn + Ln);
    for (int i = bn; i < en; ++i) rsum += Rg ? Rg[i] : r_const;
  }
  sc.tot[0 * 256 + gt] = t_card;
  sc.tot[1 * 256 + gt] = t_dwc;
  sc.tot[2 * 256 + gt] = t_dxw;
  sc.tot[3 * 256 + gt] = t_ris;
  sc.tot[4 * 256 + gt] = t_sx;
  sc.tot[5 * 256 + gt] = t_sxx;
  sc.best[gt] = rsum;
  __syncthreads();
  if (gt < 6) {  // exclusive prefix over the runs, in run order
    double acc = 0;
    for (int t = 0; t < 256; ++t) {
      const double v = sc.tot[gt * 256 + t];
      sc.tot[gt * 256 + t] = acc;
      acc += v;
    }
  } else if (gt == 6) {
    double acc = 0;
    for (int t = 0; t < 256; ++t) acc += sc.best[t];
    sc.best[256] = acc;  // total of the ranges
  }
  __syncthreads();
  const double ranges_sum = sc.best[256];
  double card = sc.tot[gt], dwc = sc.tot[256 + gt], dxw = sc.tot[512 + gt];
  double ris = ranges_sum - sc.tot[768 + gt];
  double sx = sc.tot[1024 + gt], sxx = sc.tot[1280 + gt];
  __syncthreads();
  double bcost = INFINITY, bhat = NAN;
  int bp = 0x7fffffff;
  for (int e = b; e < e_end; ++e) {
    const int tg = sc.tag[e];
    const int idx = (tg > 0 ? tg : -tg) - 1;
    const double eps = tg > 0 ? 1.0 : -1.0;
    const double x = X[idx], r = Rg ? Rg[idx] : r_const;
    const double w = 1.0 / (r * r);
    card += eps;
    dwc += eps * w;
    dxw += eps * w * x;
    ris -= eps * r;
    sx += eps * x;
    sxx += eps * x * x;
    const double x_hat = dxw / dwc;
    const double residual = card * x_hat * x_hat + sxx - 2 * sx * x_hat;
    const double cost = residual + ris;
    if (cost < bcost) {  // first minimum; NaN never wins (registration.cc:77-78)
      bcost = cost;
      bhat = x_hat;
      bp = e;
    }
  }
  sc.best[gt] = bcost;
  sc.best[257 + gt] = bhat;
  sc.bpos[gt] = bp;
  __syncthreads();
  // runs are in endpoint order, so the first strictly-smaller run wins
  double est = NAN;
  {
    double c = INFINITY;
    for (int t = 0; t < 256; ++t) {
      if (sc.best[t] < c) {
        c = sc.best[t];
        est = sc.best[257 + t];
      }
    }
    if (!(c < INFINITY)) {  // no finite cost anywhere: fall back to the first endpoint's x_hat
      est = sc.best[257];
    }
  }
  __syncthreads();
  return est;
}

constexpr int kTlsLdsEndpoints = 2048;  // per group; larger problems sort in global scratch
constexpr int kTlsGroupLds = kTlsLdsEndpoints * 12 + (6 * 256 + 514) * 8 + 256 * 4;

__device__ __forceinline__ TlsScratch tls_scratch(char* lds_group, char* glob_group, int n) {
  TlsScratch sc;
  int P2 = 2;
  while (P2 < 2 * n) P2 <<= 1;
  char* base = lds_group;
  sc.tot = reinterpret_cast<double*>(base);
  sc.best = sc.tot + 6 * 256;
  sc.bpos = reinterpret_cast<int32_t*>(sc.best + 514);
  char* ep_area = reinterpret_cast<char*>(sc.bpos + 256);
  if (P2 <= kTlsLdsEndpoints) {
    sc.val = reinterpret_cast<double*>(ep_area);
    sc.tag = reinterpret_cast<int32_t*>(sc.val + kTlsLdsEndpoints);
  } else {
    sc.val = reinterpret_cast<double*>(glob_group);
    sc.tag = reinterpret_cast<int32_t*>(sc.val + P2);
  }
  return sc;
}

// K6: translation.  One 256-thread workgroup per (axis, problem): blockIdx.x = axis a estimates
// t_a = TLS((dst - s R src)_a) (registration.cc:455-461).  Small workgroups (4 waves, 42 KB of LDS) so
// that they fit beside the K1 workgroups of the next batch; the AND of the three axis masks and
// findNonzero (registration.cc:463-470, :731) is done by whichever of the problem's three workgroups
// arrives last (agent-scope fence + counter).
// Global scratch per problem: X[3][K] doubles, mask[3][K] bytes, then 3 endpoint areas.
__global__ __launch_bounds__(256) void tls_translation_kernel(
    const ProbDesc* __restrict__ descs, const double* __restrict__ src,
    const double* __restrict__ dst, const int32_t* __restrict__ clique,
    ProbState* __restrict__ states, EstParams ep, char* __restrict__ scratch,
    int64_t scratch_stride, int32_t* __restrict__ trans_inliers) {
  TAIL_WAVE_PRIO();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int scan[257];
  __shared__ int s_last;
  const ProbDesc d = descs[blockIdx.y];
  ProbState* st = states + blockIdx.y;
  const int K = st->clique_size;
  const int g = blockIdx.x, gt = threadIdx.x;
  if (K <= 1) {
    if (g == 0 && threadIdx.x == 0) st->n_trans = 0;
    return;
  }
  char* ps = scratch + (int64_t)blockIdx.y * scratch_stride;
  const int64_t Kp = (K + 1) & ~1;
  double* X = reinterpret_cast<double*>(ps) + g * Kp;
  uint8_t* mask = reinterpret_cast<uint8_t*>(ps + 3 * Kp * 8) + (int64_t)g * Kp;
  int P2 = 2;
  while (P2 < 2 * K) P2 <<= 1;
  char* glob_ep = ps + 3 * Kp * 8 + 3 * Kp + 16 + (int64_t)g * ((int64_t)P2 * 12 + 16);
  glob_ep = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(glob_ep) + 15) & ~(uintptr_t)15);

  const double* psrc = src + 3 * d.pt_off;
  const double* pdst = dst + 3 * d.pt_off;
  const int32_t* c = clique + d.pt_off;
  const double scale = st->scale;
  // raw translation, registration.cc:726 + :455: dst - (scale * R) * src
  const double r0 = scale * st->R[3 * g], r1 = scale * st->R[3 * g + 1], r2 = scale * st->R[3 * g + 2];
  for (int j = gt; j < K; j += 256) {
    const int64_t v = c[j];
    const double a = r0 * psrc[3 * v] + r1 * psrc[3 * v + 1] + r2 * psrc[3 * v + 2];
    X[j] = pdst[3 * v + g] - a;
  }
  __syncthreads();
  const double beta = ep.noise_bound * sqrt(ep.cbar2);  // registration.cc:459 (not doubled)
  TlsScratch sc = tls_scratch(smem, glob_ep, K);
  const double est = scalar_tls_group(X, nullptr, beta, K, sc, gt);
  for (int j = gt; j < K; j += 256) mask[j] = fabs(X[j] - est) <= beta ? 1 : 0;  // :86
  if (gt == 0) st->t[g] = est;
  // publish this axis, find out whether the other two are done
  __threadfence();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-back must have landed before the counter moves
  __syncthreads();
  if (threadIdx.x == 0) s_last = (atomicAdd(&st->tls_arrive, 1) == 2) ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  __threadfence();  // acquire: the other workgroups' masks (this CU's L1 is invalidated)
  // AND of the three axis masks (registration.cc:463-470), then findNonzero (:731)
  const uint8_t* m0 = reinterpret_cast<uint8_t*>(ps + 3 * Kp * 8);
  const uint8_t* m1 = m0 + Kp;
  const uint8_t* m2 = m1 + Kp;
  const int cnt = block_compact(K, [&](int64_t j) { return (m0[j] & m1[j] & m2[j]) != 0; },
                                trans_inliers + d.pt_off, scan);
  if (threadIdx.x == 0) {
    st->n_trans = cnt;
    st->tls_arrive = 0;  // the estimators may be enqueued again on the same states (clique grew)
  }
}

void launch_tls_translation(hipStream_t s, const ProbDesc* d_desc, int batch, const double* d_src,
                            const double* d_dst, const int32_t* d_clique, ProbState* d_state,
                            EstParams ep, char* d_scratch, int64_t scratch_stride,
                            int32_t* d_trans_inliers) {
  if (batch <= 0) return;
  hipLaunchKernelGGL(tls_translation_kernel, dim3(3, batch), dim3(256), kTlsGroupLds, s, d_desc,
                     d_src, d_dst, d_clique, d_state, ep, d_scratch, scratch_stride,
                     d_trans_inliers);
}

// ------------------------------------------------------------------------------------------
// Fused estimators: ONE workgroup per problem runs the rotation stage, the translation stage, both inlier lists and
// the hand-over of the problem's state record to the host -- one launch instead of three (gnc_tls_kernel,
// tls_translation_kernel, state_push_kernel) on the serial chain of a batch.
//   FAST route (GNC-TLS, CHAIN TIMs, clique of at most kFuseMaxK vertices -- every BASELINE config):
//   * the clique's points are gathered ONCE (LDS); every thread keeps its <= 2 TIMs and weights in registers, so a
//     GNC iteration touches no memory besides the block reductions.  Sums are formed in exactly the order of
//     gnc_tls_block (thread-strided partials, xor-shuffle tree, (s0 + s1) + (s2 + s3)) and every thread runs the
//     same 3x3 SVD on the reduced H, so R, cost, iterations and weights are BIT-IDENTICAL to gnc_tls_kernel;
//   * translation (registration.cc:445-471): the three axes run on three waves in parallel.  All ranges are equal
//     (beta), so the scalar TLS of registration.cc:21-88 needs no endpoint sort with payloads: the values are sorted
//     in registers (bitonic network over the wave, no workgroup barrier), the consensus set after the e-th endpoint
//     is a contiguous WINDOW [lo, hi) of the sorted values (hi = openings so far, lo = closings so far), its sums
//     come from prefix sums, and each lane finds the merged position of its endpoints by two binary searches.
//     First minimum in endpoint order, NaN never wins -- as :77-78.  Endpoint ties (an opening and a closing of
//     exactly equal value) are unpinned in the reference (std::sort); here the opening goes first.  Window sums
//     associate differently from the sequential sweep (~1e-16 relative).
//   GENERAL route (FGR / QUATRO / COMPLETE TIMs / larger cliques): the block functions of the separate kernels,
//   the three axes one after the other.
// ------------------------------------------------------------------------------------------
constexpr int kFuseMaxK = 512;
constexpr int kFuseE = kFuseMaxK / 64;  // sorted values per lane
constexpr int kFuseLdsFast = 3 * (4 * kFuseMaxK + 8) * 8;
constexpr int kFuseLds = kFuseLdsFast > kTlsGroupLds ? kFuseLdsFast : kTlsGroupLds;

__device__ __forceinline__ double shfl_xor_d(double v, int m) { return __shfl_xor(v, m, 64); }

// ascending bitonic sort of 64 * kFuseE doubles held by one wave, blocked layout (element lane * E + r)
__device__ __forceinline__ void wave_sort_blocked(double (&v)[kFuseE], int lane) {
  constexpr int E = kFuseE, N = 64 * E;
#pragma nounroll
  for (int k = 2; k <= N; k <<= 1) {
#pragma nounroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= E) {  // partner in another lane, same register
        const int m = j / E;
        const bool upper = (lane & m) != 0;
#pragma unroll
        for (int r = 0; r < E; ++r) {
          const int idx = lane * E + r;
          const bool asc = (idx & k) == 0;
          const double o = shfl_xor_d(v[r], m);
          const double lo = v[r] < o ? v[r] : o, hi = v[r] < o ? o : v[r];
          v[r] = (asc != upper) ? lo : hi;  // the lower index of the pair keeps the smaller value when ascending
        }
      } else {  // partner in this lane: register r ^ j  (j = 1, 2, 4)
        auto inlane = [&](auto jc) {
          constexpr int JJ = decltype(jc)::value;
#pragma unroll
          for (int r = 0; r < E; ++r) {
            if ((r & JJ) == 0) {
              const int idx = lane * E + r;
              const bool asc = (idx & k) == 0;
              const double a = v[r], b = v[r | JJ];
              const double lo = a < b ? a : b, hi = a < b ? b : a;
              v[r] = asc ? lo : hi;
              v[r | JJ] = asc ? hi : lo;
            }
          }
        };
        if (j == 1) inlane(std::integral_constant<int, 1>());
        if (j == 2) inlane(std::integral_constant<int, 2>());
        if (j == 4) inlane(std::integral_constant<int, 4>());
      }
    }
  }
}
static_assert(kFuseE == 8, "wave_sort_blocked handles in-lane distances 1, 2, 4");

// ordered list of { j < count * 256 : flag(j) } for elements j = q * 256 + tid held as per-thread flags fl[q]
template <int Q>
__device__ __forceinline__ int block_compact_flags(const bool (&fl)[Q], int32_t* out, int* tab /* LDS Q * 4 + 1 */) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint64_t bal[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    bal[q] = __builtin_amdgcn_ballot_w64(fl[q]);
    if (lane == 0) tab[q * 4 + wave] = __builtin_popcountll(bal[q]);
  }
  __syncthreads();
  int total = 0;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    int base = total;
    for (int w = 0; w < 4; ++w) {
      const int cnt = tab[q * 4 + w];
      base += (w < wave) ? cnt : 0;
      total += cnt;
    }
    if (fl[q]) {
      const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(bal[q] >> 32),
                                                            __builtin_amdgcn_mbcnt_lo((unsigned int)bal[q], 0u));
      out[pos] = q * 256 + tid;
    }
  }
  __syncthreads();
  return total;
}

__global__ __launch_bounds__(256) void estimate_fused_kernel(
    const ProbDesc* __restrict__ descs, const double* __restrict__ src, const double* __restrict__ dst,
    const int32_t* __restrict__ clique, ProbState* __restrict__ states, EstParams ep,
    double* __restrict__ weights, int32_t* __restrict__ rot_inliers, const int64_t* __restrict__ tim_off,
    char* __restrict__ tls_glob, int64_t tls_stride, int32_t* __restrict__ trans_inliers,
    uint2* __restrict__ host_states /* page-locked mirror of `states`, or null */) {
  TAIL_WAVE_PRIO();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double sh[64];
  __shared__ int scan[257];
  const int prob = blockIdx.x;
  const ProbDesc d = descs[prob];
  ProbState* st = states + prob;
  const int K = st->clique_size;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double* psrc = src + 3 * d.pt_off;
  const double* pdst = dst + 3 * d.pt_off;
  const int32_t* cq = clique + d.pt_off;
  const double scale = st->scale;
  if (K <= 1) {  // registration.cc:643-647: invalid, rotation / translation untouched
    if (tid == 0) {
      st->n_rot = 0;
      st->gnc_iters = 0;
      st->n_trans = 0;
    }
  } else if (ep.algorithm == TEASER_ROT_GNC_TLS && ep.tim_graph == 0 && K <= kFuseMaxK) {
    // ---------------- fast route ----------------
    double* Pl = reinterpret_cast<double*>(smem);  // [K][6]: src xyz, dst xyz of the clique's vertices
    double* sH = sh;        // [4][9]
    double* sS = sh + 45;   // 4
    constexpr int Q = kFuseMaxK / 256;
    double px[Q][6];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int j = q * 256 + tid;
      if (j < K) {
        const int64_t v = cq[j];
        for (int r = 0; r < 3; ++r) {
          px[q][r] = psrc[3 * v + r];
          px[q][3 + r] = pdst[3 * v + r];
        }
        for (int r = 0; r < 6; ++r) Pl[6 * j + r] = px[q][r];
      }
    }
    __syncthreads();
    // CHAIN TIMs (registration.cc:657-680, :697): TIM j = vertex (j + 1) % K minus vertex j
    const double inv_scale = 1 / scale;
    double tx[Q][3], ty[Q][3], w[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int j = q * 256 + tid;
      w[q] = 1.0;
      if (j < K) {
        const int b = (j + 1 == K) ? 0 : j + 1;
        for (int r = 0; r < 3; ++r) {
          tx[q][r] = Pl[6 * b + r] - px[q][r];
          ty[q][r] = (Pl[6 * b + 3 + r] - px[q][3 + r]) * inv_scale;
        }
      } else {
        for (int r = 0; r < 3; ++r) tx[q][r] = ty[q][r] = 0;
      }
    }
    const double nb_rot = ep.noise_bound * (2 / scale);  // registration.cc:702-704
    double noise_bound_sq = nb_rot * nb_rot;             // :793-796
    if (noise_bound_sq < 1e-16) noise_bound_sq = 1e-2;
    double mu = 1, prev_cost = INFINITY, cost = INFINITY;
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    int iters = 0;
    for (int64_t it = 0; it < ep.max_iterations; ++it) {
      ++iters;
      double hh[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        if (q * 256 + tid < K) {
          for (int r = 0; r < 3; ++r) {
            const double xw = tx[q][r] * w[q];
            hh[3 * r] += xw * ty[q][0];
            hh[3 * r + 1] += xw * ty[q][1];
            hh[3 * r + 2] += xw * ty[q][2];
          }
        }
      }
      for (int k = 0; k < 9; ++k) {
        const double v = wave_sum_d(hh[k]);
        if (lane == 0) sH[wave * 9 + k] = v;
      }
      __syncthreads();
      {
        double H[9];
        for (int k = 0; k < 9; ++k) H[k] = (sH[k] + sH[9 + k]) + (sH[18 + k] + sH[27 + k]);
        svd_rot3(H, R);  // registration.cc:809 -- every thread, same input, same result
      }
      double r2[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) r2[q] = residual_sq(R, tx[q], ty[q]);
      if (it == 0) {  // registration.cc:814-825
        double mx = -INFINITY;
#pragma unroll
        for (int q = 0; q < Q; ++q)
          if (q * 256 + tid < K) mx = r2[q] > mx ? r2[q] : mx;
        mx = wave_max_d(mx);
        if (lane == 0) sS[wave] = mx;
        __syncthreads();
        const double m01 = sS[0] > sS[1] ? sS[0] : sS[1];
        const double m23 = sS[2] > sS[3] ? sS[2] : sS[3];
        const double max_residual = m01 > m23 ? m01 : m23;
        mu = 1 / (2 * max_residual / noise_bound_sq - 1);
        __syncthreads();
        if (mu <= 0) break;
      }
      const double th1 = (mu + 1) / mu * noise_bound_sq;  // registration.cc:828-829
      const double th2 = mu / (mu + 1) * noise_bound_sq;
      double c = 0;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        if (q * 256 + tid < K) {  // registration.cc:831-844
          c += w[q] * r2[q];
          double nw;
          if (r2[q] >= th1) {
            nw = 0;
          } else if (r2[q] <= th2) {
            nw = 1;
          } else {
            nw = sqrt(noise_bound_sq * mu * (mu + 1) / r2[q]) - mu;
          }
          w[q] = nw;
        }
      }
      c = wave_sum_d(c);
      __syncthreads();  // (sH / sS of this iteration have been read by everybody)
      if (lane == 0) sS[wave] = c;
      __syncthreads();
      cost = (sS[0] + sS[1]) + (sS[2] + sS[3]);
      const double cost_diff = fabs(cost - prev_cost);  // registration.cc:847-858
      mu = mu * ep.gnc_factor;
      prev_cost = cost;
      if (cost_diff < ep.cost_threshold) break;
    }
    __syncthreads();
    // rotation inliers (registration.cc:861-865, :712-716) + the weights for the getters
    {
      double* wg = weights + tim_off[prob];
      bool fl[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int j = q * 256 + tid;
        fl[q] = j < K && w[q] >= 0.5;
        if (j < K) wg[j] = w[q];
      }
      const int cnt = block_compact_flags<Q>(fl, rot_inliers + tim_off[prob], scan);
      if (tid == 0) {
        for (int k = 0; k < 9; ++k) st->R[k] = R[k];
        st->gnc_cost = cost;
        st->gnc_iters = iters;
        st->n_rot = cnt;
      }
    }
    // ---- translation: raw = dst - (scale R) src per clique vertex (registration.cc:726, :455)
    double* Xl = reinterpret_cast<double*>(smem);           // [3][kFuseMaxK]   (Pl is dead: the TIMs are in registers)
    double* Sx = Xl + 3 * kFuseMaxK;                        // [3][kFuseMaxK]   sorted
    double* S1 = Sx + 3 * kFuseMaxK;                        // [3][kFuseMaxK + 4] exclusive prefix of x
    double* S2 = S1 + 3 * (kFuseMaxK + 4);                  // [3][kFuseMaxK + 4] exclusive prefix of x^2
    double xa[Q][3];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int j = q * 256 + tid;
      for (int a = 0; a < 3; ++a) {
        const double r0 = scale * R[3 * a], r1 = scale * R[3 * a + 1], r2v = scale * R[3 * a + 2];
        const double acc = r0 * px[q][0] + r1 * px[q][1] + r2v * px[q][2];
        xa[q][a] = px[q][3 + a] - acc;
        if (j < K) Xl[a * kFuseMaxK + j] = xa[q][a];
      }
    }
    __syncthreads();
    const double beta = ep.noise_bound * sqrt(ep.cbar2);  // registration.cc:459 (not doubled)
    double* est3 = sh + 50;                                 // [3]
    if (wave < 3) {
      const int a = wave;
      const double* X = Xl + a * kFuseMaxK;
      double* sx = Sx + a * kFuseMaxK;
      double* p1 = S1 + a * (kFuseMaxK + 4);
      double* p2 = S2 + a * (kFuseMaxK + 4);
      double v[kFuseE];
#pragma unroll
      for (int r = 0; r < kFuseE; ++r) {
        const int i = lane * kFuseE + r;
        v[r] = i < K ? X[i] : INFINITY;
      }
      wave_sort_blocked(v, lane);
      // exclusive prefix sums over the sorted order: lane-local, then across the lanes
      double l1 = 0, l2 = 0;
#pragma unroll
      for (int r = 0; r < kFuseE; ++r) {
        const int i = lane * kFuseE + r;
        sx[i] = v[r];
        if (i < K) {
          l1 += v[r];
          l2 += v[r] * v[r];
        }
      }
      double i1 = l1, i2 = l2;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const double t1 = __shfl_up(i1, o, 64), t2 = __shfl_up(i2, o, 64);
        if (lane >= o) {
          i1 += t1;
          i2 += t2;
        }
      }
      double e1 = i1 - l1, e2 = i2 - l2;  // exclusive
#pragma unroll
      for (int r = 0; r < kFuseE; ++r) {
        const int i = lane * kFuseE + r;
        if (i <= K) {
          p1[i] = e1;
          p2[i] = e2;
        }
        if (i < K) {
          e1 += v[r];
          e2 += v[r] * v[r];
        }
      }
      if (lane == 63 && K == 64 * kFuseE) {
        p1[K] = e1;
        p2[K] = e2;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // every endpoint: its position in the merged order, the window after it, the cost (registration.cc:58-75)
      double bcost = INFINITY, bhat = NAN;
      int bpos = 0x7fffffff;
      auto consider = [&](int lo, int hi, int pos) {
        const double card = (double)(hi - lo);
        const double s1 = p1[hi] - p1[lo], s2 = p2[hi] - p2[lo];
        const double x_hat = s1 / card;  // (weights 1 / beta^2 cancel)
        const double residual = card * x_hat * x_hat + s2 - 2 * s1 * x_hat;
        const double c = residual + beta * (double)(K - (hi - lo));
        if (c < bcost || (c == bcost && pos < bpos)) {  // first minimum; NaN (empty window) never wins
          bcost = c;
          bhat = x_hat;
          bpos = pos;
        }
      };
#pragma unroll
      for (int r = 0; r < kFuseE; ++r) {
        const int i = lane * kFuseE + r;
        if (i < K) {
          const double av = v[r] - beta, bv = v[r] + beta;
          // closings strictly before this opening: #{ j : x_j + beta < av }
          int lo = 0, n1 = K;
          while (n1 > 0) {
            const int half = n1 >> 1;
            if (sx[lo + half] + beta < av) {
              lo += half + 1;
              n1 -= half + 1;
            } else {
              n1 = half;
            }
          }
          consider(lo, i + 1, i + lo);
          // openings up to this closing (ties: the opening first): #{ j : x_j - beta <= bv }
          int hi = 0;
          n1 = K;
          while (n1 > 0) {
            const int half = n1 >> 1;
            if (sx[hi + half] - beta <= bv) {
              hi += half + 1;
              n1 -= half + 1;
            } else {
              n1 = half;
            }
          }
          consider(i + 1, hi, i + hi);
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double oc = __shfl_xor(bcost, o, 64), oh = __shfl_xor(bhat, o, 64);
        const int op = __shfl_xor(bpos, o, 64);
        if (oc < bcost || (oc == bcost && op < bpos)) {
          bcost = oc;
          bhat = oh;
          bpos = op;
        }
      }
      if (!(bcost < INFINITY)) bhat = p1[1] - p1[0];  // no finite cost anywhere: the first endpoint's x_hat
      if (lane == 0) est3[a] = bhat;
    }
    __syncthreads();
    {
      const double e0 = est3[0], e1 = est3[1], e2 = est3[2];
      bool fl[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) {  // registration.cc:86 per axis, :463-470 AND, :731 findNonzero
        const int j = q * 256 + tid;
        fl[q] = j < K && fabs(xa[q][0] - e0) <= beta && fabs(xa[q][1] - e1) <= beta && fabs(xa[q][2] - e2) <= beta;
      }
      const int cnt = block_compact_flags<Q>(fl, trans_inliers + d.pt_off, scan);
      if (tid == 0) {
        st->t[0] = e0;
        st->t[1] = e1;
        st->t[2] = e2;
        st->n_trans = cnt;
      }
    }
  } else {
    // ---------------- general route: the block functions of the separate kernels ----------------
    TimSource ts;
    ts.ps = psrc;
    ts.pd = pdst;
    ts.c = cq;
    ts.K = K;
    ts.mode = ep.tim_graph;
    ts.inv_scale = 1 / scale;
    const int64_t KT = ep.tim_graph == 0 ? (int64_t)K : (int64_t)K * (K - 1) / 2;
    double* wg = weights + tim_off[prob];
    const double nb = ep.noise_bound * (2 / scale);  // registration.cc:702-704
    rotation_block(ep.algorithm, ts, KT, nb, ep, wg, st->R, &st->gnc_cost, &st->gnc_iters, sh);
    const int alg = ep.algorithm;
    const int cnt = block_compact(KT, [&](int64_t j) { return rotation_inlier(alg, wg[j]); },
                                  rot_inliers + tim_off[prob], scan);
    if (tid == 0) st->n_rot = cnt;
    __threadfence_block();
    __syncthreads();
    char* ps = tls_glob + (int64_t)prob * tls_stride;  // layout as tls_translation_kernel
    const int64_t Kp = (K + 1) & ~1;
    int P2 = 2;
    while (P2 < 2 * K) P2 <<= 1;
    const double beta = ep.noise_bound * sqrt(ep.cbar2);
    for (int g = 0; g < 3; ++g) {
      double* X = reinterpret_cast<double*>(ps) + g * Kp;
      uint8_t* mask = reinterpret_cast<uint8_t*>(ps + 3 * Kp * 8) + (int64_t)g * Kp;
      char* glob_ep = ps + 3 * Kp * 8 + 3 * Kp + 16 + (int64_t)g * ((int64_t)P2 * 12 + 16);
      glob_ep = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(glob_ep) + 15) & ~(uintptr_t)15);
      const double r0 = scale * st->R[3 * g], r1 = scale * st->R[3 * g + 1], r2 = scale * st->R[3 * g + 2];
      for (int j = tid; j < K; j += 256) {
        const int64_t v = cq[j];
        X[j] = pdst[3 * v + g] - (r0 * psrc[3 * v] + r1 * psrc[3 * v + 1] + r2 * psrc[3 * v + 2]);
      }
      __syncthreads();
      TlsScratch sc = tls_scratch(smem, glob_ep, K);
      const double est = scalar_tls_group(X, nullptr, beta, K, sc, tid);
      for (int j = tid; j < K; j += 256) mask[j] = fabs(X[j] - est) <= beta ? 1 : 0;
      if (tid == 0) st->t[g] = est;
      __syncthreads();
    }
    const uint8_t* m0 = reinterpret_cast<uint8_t*>(ps + 3 * Kp * 8);
    const uint8_t* m1 = m0 + Kp;
    const uint8_t* m2 = m1 + Kp;
    const int cnt2 = block_compact(K, [&](int64_t j) { return (m0[j] & m1[j] & m2[j]) != 0; },
                                   trans_inliers + d.pt_off, scan);
    if (tid == 0) st->n_trans = cnt2;
  }
  // ---- the problem's state record goes to the host mirror (what state_push_kernel did for the whole batch)
  if (host_states) {
    // (the record was written by this workgroup's threads through the L1 of this CU: release, meet, acquire at
    // agent scope -- the acquire drops the CU's possibly stale lines -- before other threads read it back)
    __threadfence();
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    constexpr int n8 = (int)(sizeof(ProbState) / 8);
    const uint2* from = reinterpret_cast<const uint2*>(st);
    uint2* to = host_states + (size_t)prob * n8;
    if (tid < n8) to[tid] = from[tid];
  }
}

void launch_estimate_fused(hipStream_t s, const ProbDesc* d_desc, int batch, const double* d_src, const double* d_dst,
                           const int32_t* d_clique, ProbState* d_state, EstParams ep, double* d_weights,
                           int32_t* d_rot_inliers, const int64_t* d_tim_off, char* d_tls_scratch, int64_t tls_stride,
                           int32_t* d_trans_inliers, void* host_states) {
  if (batch <= 0) return;
  hipLaunchKernelGGL(estimate_fused_kernel, dim3(batch), dim3(256), kFuseLds, s, d_desc, d_src, d_dst, d_clique,
                     d_state, ep, d_weights, d_rot_inliers, d_tim_off, d_tls_scratch, tls_stride, d_trans_inliers,
                     reinterpret_cast<uint2*>(host_states));
}

// Stand-alone scalar TLS on device arrays x[n], r[n] (one group).
__global__ __launch_bounds__(256) void scalar_tls_kernel(const double* __restrict__ x,
                                                         const double* __restrict__ r, int n,
                                                         char* __restrict__ scratch,
                                                         double* __restrict__ est_out,
                                                         uint8_t* __restrict__ mask) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TlsScratch sc = tls_scratch(smem, scratch, n);
  const double est = scalar_tls_group(x, r, 0.0, n, sc, threadIdx.x);
  if (mask)
    for (int j = threadIdx.x; j < n; j += 256) mask[j] = fabs(x[j] - est) <= r[j] ? 1 : 0;
  if (threadIdx.x == 0) *est_out = est;
}

// ------------------------------------------------------------------------------------------
// estimate_scaling = true for a BATCH of small problems (n <= 724: at most 2^19 interval endpoints, sorted by
// one workgroup each): every problem's TRIMs in one launch, every problem's scalar TLS in another -- instead
// of two launches per problem one after the other (each single-workgroup sort leaves 255 CUs idle).
// sel[k] = problem index; off[2k] = offset (in doubles) of problem k's TRIM arrays, off[2k+1] = byte offset of
// its sort scratch.  Same arithmetic and pair order as trims_kernel / scalar_tls_kernel.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void trims_batch_kernel(const ProbDesc* __restrict__ descs,
                                                          const int32_t* __restrict__ sel,
                                                          const int64_t* __restrict__ off,
                                                          const double* __restrict__ src,
                                                          const double* __restrict__ dst, double beta,
                                                          double* __restrict__ raw, double* __restrict__ alpha) {
  const ProbDesc d = descs[sel[blockIdx.y]];
  const int n = d.n, i = blockIdx.x;
  if (i >= n - 1) return;
  const double* ps = src + 3 * d.pt_off;
  const double* pd = dst + 3 * d.pt_off;
  double* rw = raw + off[2 * blockIdx.y];
  double* al = alpha + off[2 * blockIdx.y];
  const int64_t seg = (int64_t)i * n - (int64_t)i * (i + 1) / 2;
  const double six = ps[3 * i], siy = ps[3 * i + 1], siz = ps[3 * i + 2];
  const double dix = pd[3 * i], diy = pd[3 * i + 1], diz = pd[3 * i + 2];
  for (int j = i + 1 + threadIdx.x; j < n; j += 256) {
    const double ax = ps[3 * j] - six, ay = ps[3 * j + 1] - siy, az = ps[3 * j + 2] - siz;
    const double bx = pd[3 * j] - dix, by = pd[3 * j + 1] - diy, bz = pd[3 * j + 2] - diz;
    const double v1 = __builtin_sqrt((ax * ax + ay * ay) + az * az);
    const double v2 = __builtin_sqrt((bx * bx + by * by) + bz * bz);
    const int64_t k = seg + (j - i - 1);
    rw[k] = v2 / v1;
    al[k] = beta * (1.0 / v1);
  }
}

__global__ __launch_bounds__(256) void scalar_tls_batch_kernel(const ProbDesc* __restrict__ descs,
                                                               const int32_t* __restrict__ sel,
                                                               const int64_t* __restrict__ off,
                                                               const double* __restrict__ raw,
                                                               const double* __restrict__ alpha,
                                                               char* __restrict__ scratch,
                                                               ProbState* __restrict__ states) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int p = sel[blockIdx.x];
  const int n = descs[p].n;
  const int m = (int)((int64_t)n * (n - 1) / 2);
  TlsScratch sc = tls_scratch(smem, scratch + off[2 * blockIdx.x + 1], m);
  const double est = scalar_tls_group(raw + off[2 * blockIdx.x], alpha + off[2 * blockIdx.x], 0.0, m, sc, threadIdx.x);
  if (threadIdx.x == 0) states[p].scale = est;
}

void launch_scale_small_batch(hipStream_t s, const ProbDesc* d_desc, const int32_t* d_sel, const int64_t* d_off,
                              int count, int max_n, const double* d_src, const double* d_dst, double beta,
                              double* d_raw, double* d_alpha, char* d_scratch, ProbState* d_state) {
  if (count <= 0 || max_n < 2) return;
  hipLaunchKernelGGL(trims_batch_kernel, dim3(max_n - 1, count), dim3(256), 0, s, d_desc, d_sel, d_off, d_src,
                     d_dst, beta, d_raw, d_alpha);
  hipLaunchKernelGGL(scalar_tls_batch_kernel, dim3(count), dim3(256), kTlsGroupLds, s, d_desc, d_sel, d_off, d_raw,
                     d_alpha, d_scratch, d_state);
}

void launch_scalar_tls(hipStream_t s, const double* d_x, const double* d_r, int32_t n,
                       char* d_scratch, double* d_est, uint8_t* d_mask) {
  hipLaunchKernelGGL(scalar_tls_kernel, dim3(1), dim3(256), kTlsGroupLds, s, d_x, d_r, n, d_scratch,
                     d_est, d_mask);
}

}  // namespace thip
