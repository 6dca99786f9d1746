// cert_setup.h -- host-side O(N) set-up of the DRS certifier (plain C++, no HIP: also compiled by
// tests/test_certifier_setup.py against oracle/certifier.py): from (R, src, dst, theta) the 4 x 4 blocks of
//   M_init = D_omega^T Q_cost D_omega - mu J - lambda_guess      (certification.cc:58-98)
// which is non-zero only in its first block row / column and its diagonal blocks, plus mu = x^T Q_cost x.
// Blocks are column-major 4 x 4: diag[(N+1)][16], row0[N][16] = block (0, k+1), col0[N][16] = block (k+1, 0).
// src / dst: N points, xyz interleaved (the reference's 3 x N column-major matrices); R row-major 3 x 3.
#pragma once

#include <cmath>
#include <cstring>
#include <vector>

namespace thip {

inline void cert_mat4_mul(const double* A, const double* B, double* C) {  // column-major 4 x 4
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += A[k * 4 + r] * B[c * 4 + k];
      C[c * 4 + r] = s;
    }
}

// Eigen::Quaterniond(R).normalize() (certification.cc:67-68) -> (x, y, z, w)
inline void cert_rot_to_quat(const double* R, double* q) {
  auto at = [&](int r, int c) { return R[3 * r + c]; };
  double t = at(0, 0) + at(1, 1) + at(2, 2);
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (at(2, 1) - at(1, 2)) * t;
    q[1] = (at(0, 2) - at(2, 0)) * t;
    q[2] = (at(1, 0) - at(0, 1)) * t;
  } else {
    int i = 0;
    if (at(1, 1) > at(0, 0)) i = 1;
    if (at(2, 2) > at(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    t = std::sqrt(at(i, i) - at(j, j) - at(k, k) + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (at(k, j) - at(j, k)) * t;
    q[j] = (at(j, i) + at(i, j)) * t;
    q[k] = (at(k, i) + at(i, k)) * t;
  }
  const double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; ++k) q[k] /= nrm;
}

inline void cert_setup(const double* R, const double* src, const double* dst, const double* theta, int N,
                       double noise_bound, double cbar2, std::vector<double>* thp_out, std::vector<double>* diag_out,
                       std::vector<double>* row0_out, std::vector<double>* col0_out, double* mu_out) {
  // coefficient matrix that maps vec(q q^T) to vec(R) (certification.cc:241-252)
  static const double kP[9][16] = {
      {1, 0, 0, 0, 0, -1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1},  {0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0},
      {0, 0, 1, 0, 0, 0, 0, -1, 1, 0, 0, 0, 0, -1, 0, 0},  {0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0, -1, 0, 0, -1, 0},
      {-1, 0, 0, 0, 0, 1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1},  {0, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 0},
      {0, 0, 1, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 1, 0, 0},    {0, 0, 0, -1, 0, 0, 1, 0, 0, 1, 0, 0, -1, 0, 0, 0},
      {-1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
  const double nbs = cbar2 * noise_bound * noise_bound;
  double q[4];
  cert_rot_to_quat(R, q);
  // getOmega1 (certification.cc:301-310), column-major, and its transpose
  const double om[16] = {q[3], q[2], -q[1], -q[0], -q[2], q[3], q[0], -q[1],
                         q[1], -q[0], q[3], -q[2], q[0], q[1], q[2], q[3]};
  double omT[16];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) omT[c * 4 + r] = om[r * 4 + c];
  std::vector<double>& thp = *thp_out;
  std::vector<double>& diag = *diag_out;
  std::vector<double>& row0 = *row0_out;
  std::vector<double>& col0 = *col0_out;
  thp.assign((size_t)N + 1, 1.0);
  for (int k = 0; k < N; ++k) thp[(size_t)k + 1] = theta[k];
  diag.assign((size_t)(N + 1) * 16, 0.0);
  row0.assign((size_t)N * 16, 0.0);
  col0.assign((size_t)N * 16, 0.0);
  std::vector<double> qd((size_t)(N + 1) * 16, 0.0), q0((size_t)N * 16);  // Q_cost: diagonal blocks, (0,k) = (k,0)
  for (int k = 0; k < N; ++k) {  // getQCost (certification.cc:233-299)
    const double* a = src + 3 * k;
    const double* b = dst + 3 * k;
    double A9[9];  // v2 v1^T, column-major
    for (int c = 0; c < 3; ++c)
      for (int r = 0; r < 3; ++r) A9[c * 3 + r] = b[r] * a[c];
    double Pk[16];
    for (int e = 0; e < 16; ++e) {
      double s = 0;
      for (int m = 0; m < 9; ++m) s += kP[m][e] * A9[m];
      Pk[e] = s;  // Eigen::Map<Matrix4d> of P^T vec(v2 v1^T): column-major
    }
    const double nn = (a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) + (b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    const double ck1 = 0.5 * (nn - nbs), ck2 = 0.5 * (nn + nbs);
    for (int e = 0; e < 16; ++e) {
      const double id = (e % 5 == 0) ? 1.0 : 0.0;
      q0[(size_t)k * 16 + e] = -0.5 * Pk[e] + ck1 / 2 * id;
      qd[(size_t)(k + 1) * 16 + e] = -Pk[e] + ck2 * id;
    }
  }
  // mu = x^T Q_cost x with x = kron(theta_prepended, q) (certification.cc:73-92)
  double mu = 0;
  auto quad = [&](const double* Bm) {  // q^T B q
    double s = 0;
    for (int c = 0; c < 4; ++c)
      for (int r = 0; r < 4; ++r) s += q[r] * Bm[c * 4 + r] * q[c];
    return s;
  };
  for (int k = 0; k < N; ++k) {
    const double t = thp[(size_t)k + 1];
    mu += 2.0 * t * quad(&q0[(size_t)k * 16]);  // blocks (0, k+1) and (k+1, 0); theta_prepended(0) = 1
    mu += t * t * quad(&qd[(size_t)(k + 1) * 16]);
  }
  auto conj = [&](const double* Bm, double* out) {  // Omega^T B Omega
    double t[16];
    cert_mat4_mul(omT, Bm, t);
    cert_mat4_mul(t, om, out);
  };
  double top[16] = {0};
  for (int k = 0; k < N; ++k) {
    conj(&q0[(size_t)k * 16], &row0[(size_t)k * 16]);
    memcpy(&col0[(size_t)k * 16], &row0[(size_t)k * 16], 16 * sizeof(double));
    double qb[16];
    conj(&qd[(size_t)(k + 1) * 16], qb);
    // getLambdaGuess (certification.cc:454-536): the block of correspondence k
    const double* sv = src + 3 * k;
    const double* dv = dst + 3 * k;
    double d[3], xi[3];
    for (int r = 0; r < 3; ++r) d[r] = dv[r] - (R[3 * r] * sv[0] + R[3 * r + 1] * sv[1] + R[3 * r + 2] * sv[2]);
    for (int r = 0; r < 3; ++r) xi[r] = R[r] * d[0] + R[3 + r] * d[1] + R[6 + r] * d[2];  // R^T (dst - R src)
    const double n2 = xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2];
    const double sxi = sv[0] * xi[0] + sv[1] * xi[1] + sv[2] * xi[2];
    const double sh[9] = {0, sv[2], -sv[1], -sv[2], 0, sv[0], sv[1], -sv[0], 0};  // hatmap, column-major
    const double xh[9] = {0, xi[2], -xi[1], -xi[2], 0, xi[0], xi[1], -xi[0], 0};
    auto m3 = [](const double* A, const double* B, double* C) {
      for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) C[c * 3 + r] = A[r] * B[c * 3] + A[3 + r] * B[c * 3 + 1] + A[6 + r] * B[c * 3 + 2];
    };
    double shsh[9], xhsh[9];
    m3(sh, sh, shsh);
    m3(xh, sh, xhsh);
    const bool pos = theta[k] > 0;
    const double cn = pos ? 0.75 : 0.25, cv = pos ? -1.5 : -0.5;
    double cur[16] = {0};
    cur[15] = pos ? (-0.75 * n2 - 0.25 * nbs) : (-0.25 * n2 - 0.75 * nbs);
    for (int c = 0; c < 3; ++c)
      for (int r = 0; r < 3; ++r) {
        const double id = (r == c) ? 1.0 : 0.0;
        cur[c * 4 + r] = shsh[c * 3 + r] - 0.5 * sxi * id + 0.5 * xhsh[c * 3 + r] + 0.5 * xi[r] * sv[c] - cn * n2 * id -
                         0.25 * nbs * id;
      }
    for (int r = 0; r < 3; ++r) {
      const double v = cv * (xh[r] * sv[0] + xh[3 + r] * sv[1] + xh[6 + r] * sv[2]);
      cur[12 + r] = v;     // column 3
      cur[r * 4 + 3] = v;  // row 3
    }
    for (int e = 0; e < 16; ++e) {
      diag[(size_t)(k + 1) * 16 + e] = qb[e] + cur[e];  // Q_bar - lambda (the lambda block is -cur)
      top[e] += cur[e];
    }
  }
  for (int e = 0; e < 16; ++e) diag[(size_t)e] = -mu * ((e % 5 == 0) ? 1.0 : 0.0) - top[e];  // Q_bar(0,0) = 0
  *mu_out = mu;
}

}  // namespace thip
