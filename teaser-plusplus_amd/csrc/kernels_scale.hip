// kernels_scale.hip -- scalar TLS over MANY measurements (the estimate_scaling = true stage for
// any n): TLSScaleSolver::solveForScale (reference teaser/src/registration.cc:410-425) feeding
// ScalarTLSEstimator::estimate (registration.cc:21-88) with M = n(n-1)/2 TRIMs.
//
//   1. trim_endpoints_kernel   raw_k = |b_k|/|a_k|, alpha_k = beta/|a_k| for every pair in the
//                              reference's pair order (registration.cc:531) and, fused, the 2M
//                              interval endpoints (value, tag = +-(k+1)) of registration.cc:35-38
//   2. device radix sort       ascending by value only (registration.cc:41-42).  The sort is stable
//                              and the endpoints are generated in insertion order, so ties resolve
//                              exactly like the single-workgroup path and the oracle's merge sort.
//   3. the sweep of registration.cc:58-75 as a three-pass blocked prefix sum over the sorted
//      endpoints (chunk totals -> exclusive scan of the totals -> per-endpoint cost + arg-min),
//      first minimum wins, NaN never wins (registration.cc:77-78).
// The running sums are therefore associated differently from the reference's sequential loop
// (~1e-16 relative); the estimate is compared against the oracle at 1e-9.
#include <cstring>
#include <cstdint>

#include <rocprim/device/device_radix_sort.hpp>

#include "internal.h"

namespace thip {

namespace {

constexpr int kSwThreads = 256;
constexpr int kSwPer = 8;                         // consecutive endpoints per thread
constexpr int kSwChunk = kSwThreads * kSwPer;     // endpoints per workgroup
constexpr int kNumAcc = 7;                        // card, dwc, dxw, ris, sx, sxx, opening ranges

struct Acc {
  double v[kNumAcc];
};

// measurement behind an endpoint tag: (x, range) from two arrays, or from ONE interleaved array when r == nullptr
// (the TRIM kernels write (raw, alpha) pairs: one 16-byte gather per endpoint instead of two 8-byte ones)
__device__ __forceinline__ double2 fetch_xr(int tag, const double* __restrict__ x, const double* __restrict__ r) {
  const int idx = (tag > 0 ? tag : -tag) - 1;
  if (r) return make_double2(x[idx], r[idx]);
  return reinterpret_cast<const double2*>(x)[idx];
}
__device__ __forceinline__ void acc_add(Acc& a, int tag, double xv, double rv) {
  const double eps = tag > 0 ? 1.0 : -1.0;
  const double w = 1.0 / (rv * rv);  // weights = ranges^-2, registration.cc:45-46
  a.v[0] += eps;
  a.v[1] += eps * w;
  a.v[2] += eps * w * xv;
  a.v[3] += eps * rv;
  a.v[4] += eps * xv;
  a.v[5] += eps * xv * xv;
  a.v[6] += tag > 0 ? rv : 0.0;
}

__device__ __forceinline__ double shfl_up_d(double v, int d) { return __shfl_up(v, d, 64); }
__device__ __forceinline__ double shfl_down_d(double v, int d) { return __shfl_down(v, d, 64); }

// The two endpoint keys of measurement k: FP64 for the 64-bit sort, or ROUNDED TO FLOAT for the 32-bit sort of
// the float-key path (below: the exact order is restored inside runs of equal float keys).
__device__ __forceinline__ void store_endpoint_keys(double* __restrict__ keys, float* __restrict__ fkeys, int64_t k,
                                                    double lo, double hi) {
  if (fkeys)
    *reinterpret_cast<float2*>(fkeys + 2 * k) = make_float2((float)lo, (float)hi);
  else
    *reinterpret_cast<double2*>(keys + 2 * k) = make_double2(lo, hi);
}

// TRIM k of a problem: raw scale s = |b| / |a| and range alpha = beta / |a| of the pair (i, j), i < j
// (registration.cc:415-422).  ONE definition: the endpoint kernels derive the sort keys from it and the order-fix
// kernel recomputes the same two doubles from the tag instead of gathering them from an 800 MB array.
__device__ __forceinline__ void trim_terms(const double* __restrict__ src, const double* __restrict__ dst, int i, int j,
                                           double beta, double* s, double* a) {
  const double ax = src[3 * j] - src[3 * i], ay = src[3 * j + 1] - src[3 * i + 1], az = src[3 * j + 2] - src[3 * i + 2];
  const double bx = dst[3 * j] - dst[3 * i], by = dst[3 * j + 1] - dst[3 * i + 1], bz = dst[3 * j + 2] - dst[3 * i + 2];
  const double v1 = __builtin_sqrt((ax * ax + ay * ay) + az * az);  // registration.cc:415-418
  const double v2 = __builtin_sqrt((bx * bx + by * by) + bz * bz);
  *s = v2 / v1;              // registration.cc:420
  *a = beta * (1.0 / v1);    // registration.cc:422
}
// inverse of the reference's pair order k = i n - i (i + 1) / 2 + (j - i - 1)  (registration.cc:531)
__device__ __forceinline__ void trim_pair_of(int64_t k, int n, int* i, int* j) {
  const double t = 2.0 * n - 1.0;
  int r = (int)((t - __builtin_sqrt(t * t - 8.0 * (double)k)) * 0.5);
  r = r < 0 ? 0 : (r > n - 2 ? n - 2 : r);
  while (r > 0 && (int64_t)r * n - (int64_t)r * (r + 1) / 2 > k) --r;
  while (r < n - 2 && (int64_t)(r + 1) * n - (int64_t)(r + 1) * (r + 2) / 2 <= k) ++r;
  *i = r;
  *j = (int)(k - ((int64_t)r * n - (int64_t)r * (r + 1) / 2)) + r + 1;
}

// unsigned integers that order like the floats they encode (-0.0f < +0.0f, as the radix sort of floats has it)
__device__ __forceinline__ unsigned int float_order_bits(float v) {
  const unsigned int b = __float_as_uint(v);
  return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}

// ---- float-key path: 4 radix passes over 8-byte items instead of 8 over 12-byte ones ---------------------
// double -> float rounding is monotone, so a STABLE sort by the float key leaves the endpoints in the exact
// order up to permutations inside runs of equal float keys (2^-24 relative resolution: runs of a few items at
// 1e8 endpoints), and inside a run the items are still in insertion order.  This kernel restores the exact
// order: every item recomputes its FP64 key from the measurement behind its tag (the gather the sweep needs
// anyway: the measurements leave in sorted order for passes A and C), ranks itself inside its run -- key first,
// position second, i.e. exactly what the stable 64-bit sort produces -- and moves to its final slot.  A run that
// reaches more than kFxHalo items beyond a chunk (degenerate data: thousands of equal measurements) raises the
// overflow flag; the host then repeats the stage with the 64-bit sort.
constexpr int kFxHalo = 64;
__global__ __launch_bounds__(kSwThreads) void tls_order_fix_kernel(
    const uint32_t* __restrict__ fk, int fk_stride, const int32_t* __restrict__ tags, const double* __restrict__ x,
    const double* __restrict__ r, int64_t m, int64_t nblk, int32_t* __restrict__ out_tags,
    double2* __restrict__ out_xr, int32_t* __restrict__ overflow, int64_t overflow_stride,
    const ScaleSeg* __restrict__ segs, const double* __restrict__ src, const double* __restrict__ dst, int n_pts,
    double beta) {
  __shared__ uint32_t s_fk[kSwChunk + 2 * kFxHalo];
  __shared__ double s_kd[kSwChunk + 2 * kFxHalo];
  int64_t trim0 = 0;  // first TRIM of this problem (tags are global in a batch)
  if (segs) {
    const ScaleSeg sg = segs[blockIdx.y];
    if ((int64_t)blockIdx.x >= sg.nblk) return;
    fk += sg.e_off * fk_stride;
    tags += sg.e_off;
    out_tags += sg.e_off;
    out_xr += sg.e_off;
    m = sg.m;
    overflow = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(overflow) + (int64_t)sg.prob * overflow_stride);
    trim0 = sg.trim_off;
    n_pts = sg.n;
    if (src) {
      src += 3 * sg.pt_off;
      dst += 3 * sg.pt_off;
    }
  }
  // the measurement behind a tag: recomputed from the points (TRIM paths: nothing is stored per TRIM -- measured,
  // the twelve scattered cache-resident loads + two sqrt + two divisions cost what the random 16-byte HBM gather
  // of a stored (raw, alpha) pair costs, 1.4 - 1.6 of this kernel's 2.7 ms at 1e8 endpoints, without the 800 MB
  // array) or gathered (caller-supplied measurements)
  auto measure = [&](int tag) -> double2 {
    if (!src) return fetch_xr(tag, x, r);
    int i, j;
    trim_pair_of((int64_t)(tag > 0 ? tag : -tag) - 1 - trim0, n_pts, &i, &j);
    double sv, av;
    trim_terms(src, dst, i, j, beta, &sv, &av);
    return make_double2(sv, av);
  };
  const int64_t c0 = (int64_t)blockIdx.x * kSwChunk;
  const int64_t lo = c0 - kFxHalo;  // LDS index i <-> position lo + i
  int tg[kSwPer];
  double2 xr[kSwPer];
  for (int e = 0; e < kSwPer; ++e) {
    const int64_t pos = c0 + threadIdx.x + (int64_t)e * kSwThreads;
    tg[e] = pos < m ? tags[pos] : 0;
  }
  for (int e = 0; e < kSwPer; ++e)
    if (tg[e] != 0) xr[e] = measure(tg[e]);
  for (int e = 0; e < kSwPer; ++e) {
    const int li = kFxHalo + threadIdx.x + e * kSwThreads;
    const int64_t pos = lo + li;
    if (tg[e] != 0) {
      s_fk[li] = fk[pos * fk_stride];
      s_kd[li] = tg[e] > 0 ? xr[e].x - xr[e].y : xr[e].x + xr[e].y;  // the endpoint kernels' s - a / s + a
    }
  }
  if (threadIdx.x < 2 * kFxHalo) {  // the halos
    const int li = threadIdx.x < kFxHalo ? threadIdx.x : kSwChunk + threadIdx.x;
    const int64_t pos = lo + li;
    if (pos >= 0 && pos < m) {
      const int t = tags[pos];
      const double2 v = measure(t);
      s_fk[li] = fk[pos * fk_stride];
      s_kd[li] = t > 0 ? v.x - v.y : v.x + v.y;
    }
  }
  __syncthreads();
  bool over = false;
  for (int e = 0; e < kSwPer; ++e) {
    if (tg[e] == 0) continue;
    const int li = kFxHalo + threadIdx.x + e * kSwThreads;
    const int64_t pos = lo + li;
    const uint32_t f = s_fk[li];
    const double kd = s_kd[li];
    int shift = 0;
    // items of the run in front of this one that must end up behind it (strictly larger key) ...
    int j = li - 1;
    for (; j >= 0 && lo + j >= 0 && s_fk[j] == f; --j) shift -= s_kd[j] > kd ? 1 : 0;
    if (j < 0 && lo + j >= 0) over = true;  // the run continues beyond the halo
    // ... and behind it that must end up in front (strictly smaller key)
    j = li + 1;
    for (; j < kSwChunk + 2 * kFxHalo && lo + j < m && s_fk[j] == f; ++j) shift += s_kd[j] < kd ? 1 : 0;
    if (j >= kSwChunk + 2 * kFxHalo && lo + j < m) over = true;
    out_tags[pos + shift] = tg[e];
    out_xr[pos + shift] = xr[e];
  }
  if (over) *overflow = 1;
}

// One row of pairs per workgroup (row i, columns j > i), like trims_kernel, plus the endpoints.
__global__ __launch_bounds__(256) void trim_endpoints_kernel(
    const double* __restrict__ src, const double* __restrict__ dst, int n, double beta,
    double* __restrict__ raw, double* __restrict__ alpha, double* __restrict__ keys,
    int32_t* __restrict__ tags, float* __restrict__ fkeys) {
  const int i = blockIdx.x;
  if (i >= n - 1) return;
  const int64_t seg = (int64_t)i * n - (int64_t)i * (i + 1) / 2;
  for (int j = i + 1 + threadIdx.x; j < n; j += 256) {
    double s, a;
    trim_terms(src, dst, i, j, beta, &s, &a);
    const int64_t k = seg + (j - i - 1);
    // (float-key path: the order-fix kernel recomputes the pair, nothing is kept per TRIM)
    if (!fkeys) reinterpret_cast<double2*>(raw)[k] = make_double2(s, a);  // interleaved (alpha unused)
    // registration.cc:35-38
    store_endpoint_keys(keys, fkeys, k, s - a, s + a);
    *reinterpret_cast<int2*>(tags + 2 * k) = make_int2((int)(k + 1), -(int)(k + 1));
  }
}

// The same for a whole batch of problems (grid (max n - 1, problems)): problem q's TRIMs live at
// [trim_off, trim_off + M_q) of raw / alpha, its endpoints at twice that, and the tags are GLOBAL
// (+-(trim_off + k + 1)), so one device-wide sort can carry all the problems at once.
__global__ __launch_bounds__(256) void trim_endpoints_batch_kernel(
    const double* __restrict__ src_all, const double* __restrict__ dst_all, const ScaleSeg* __restrict__ segs,
    double beta, double* __restrict__ raw, double* __restrict__ alpha, double* __restrict__ keys,
    int32_t* __restrict__ tags, unsigned long long* __restrict__ ckeys) {
  const ScaleSeg sg = segs[blockIdx.y];
  const int n = sg.n, i = blockIdx.x;
  if (i >= n - 1) return;
  const double* src = src_all + 3 * sg.pt_off;
  const double* dst = dst_all + 3 * sg.pt_off;
  const int64_t seg = sg.trim_off + (int64_t)i * n - (int64_t)i * (i + 1) / 2;
  for (int j = i + 1 + threadIdx.x; j < n; j += 256) {
    double sc, a;
    trim_terms(src, dst, i, j, beta, &sc, &a);
    const int64_t k = seg + (j - i - 1);
    if (ckeys) {
      // float-key path of a batch: ONE sort on (problem slot, float key) -- the slot in the high word, the float's
      // order-preserving bit pattern in the low word
      const unsigned long long hi = (unsigned long long)blockIdx.y << 32;
      *reinterpret_cast<ulonglong2*>(ckeys + 2 * k) =
          make_ulonglong2(hi | float_order_bits((float)(sc - a)), hi | float_order_bits((float)(sc + a)));
    } else {
      reinterpret_cast<double2*>(raw)[k] = make_double2(sc, a);  // interleaved (alpha unused)
      store_endpoint_keys(keys, nullptr, k, sc - a, sc + a);
    }
    *reinterpret_cast<int2*>(tags + 2 * k) = make_int2((int)(k + 1), -(int)(k + 1));
  }
}

// problem slot of every endpoint after the value sort (binary search of its TRIM index in the segment table):
// the key of the second, stable, pass that gathers the problems back into contiguous segments
__global__ __launch_bounds__(256) void scale_slot_kernel(const int32_t* __restrict__ tags, int64_t m,
                                                         const ScaleSeg* __restrict__ segs, int count,
                                                         uint32_t* __restrict__ slot) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= m) return;
  const int tag = tags[e];
  const int64_t k = (tag > 0 ? tag : -tag) - 1;
  int lo = 0, hi = count - 1;
  while (lo < hi) {  // the last segment with trim_off <= k
    const int mid = (lo + hi + 1) >> 1;
    if (segs[mid].trim_off <= k)
      lo = mid;
    else
      hi = mid - 1;
  }
  slot[e] = (uint32_t)lo;
}

__global__ __launch_bounds__(256) void tls_endpoints_kernel(const double* __restrict__ x,
                                                            const double* __restrict__ r, int64_t n,
                                                            double* __restrict__ keys,
                                                            int32_t* __restrict__ tags,
                                                            float* __restrict__ fkeys) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const double s = x[k], a = r[k];
  store_endpoint_keys(keys, fkeys, k, s - a, s + a);
  *reinterpret_cast<int2*>(tags + 2 * k) = make_int2((int)(k + 1), -(int)(k + 1));
}

// pass A: totals of every chunk of kSwChunk sorted endpoints.  partials is SoA [kNumAcc][nblk].
__global__ __launch_bounds__(kSwThreads) void tls_sweep_totals_kernel(
    const int32_t* __restrict__ tags, const double* __restrict__ x, const double* __restrict__ r,
    int64_t m, int64_t nblk, double* __restrict__ partials, double2* __restrict__ sorted_xr,
    const ScaleSeg* __restrict__ segs) {
  __shared__ double red[kSwThreads / 64][kNumAcc];
  if (segs) {  // batch: blockIdx.y = problem slot, everything relative to its segment
    const ScaleSeg sg = segs[blockIdx.y];
    if ((int64_t)blockIdx.x >= sg.nblk) return;
    tags += sg.e_off;
    sorted_xr += sg.e_off;
    partials += kNumAcc * sg.blk_off;
    m = sg.m;
    nblk = sg.nblk;
  }
  const int64_t base = (int64_t)blockIdx.x * kSwChunk + (int64_t)threadIdx.x * kSwPer;
  Acc a;
  for (int q = 0; q < kNumAcc; ++q) a.v[q] = 0;
  // the ONLY random gather of the sweep: the measurements are left behind in sorted order for pass C
  int tg[kSwPer];
  double2 xr[kSwPer];
  for (int e = 0; e < kSwPer; ++e) tg[e] = base + e < m ? tags[base + e] : 0;
  if (x == nullptr) {  // float-key path: the order-fix kernel has left the measurements in sorted order
    for (int e = 0; e < kSwPer; ++e)
      if (tg[e] != 0) xr[e] = sorted_xr[base + e];
    for (int e = 0; e < kSwPer; ++e)
      if (tg[e] != 0) acc_add(a, tg[e], xr[e].x, xr[e].y);
  } else {
    for (int e = 0; e < kSwPer; ++e)
      if (tg[e] != 0) xr[e] = fetch_xr(tg[e], x, r);
    for (int e = 0; e < kSwPer; ++e)
      if (tg[e] != 0) {
        sorted_xr[base + e] = xr[e];
        acc_add(a, tg[e], xr[e].x, xr[e].y);
      }
  }
  // fixed-shape tree inside the wave, then the waves in order
  for (int q = 0; q < kNumAcc; ++q) {
    double v = a.v[q];
    for (int off = 32; off > 0; off >>= 1) v += shfl_down_d(v, off);
    a.v[q] = v;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
    for (int q = 0; q < kNumAcc; ++q) red[wave][q] = a.v[q];
  __syncthreads();
  if (threadIdx.x < kNumAcc) {
    double v = 0;
    for (int w = 0; w < kSwThreads / 64; ++w) v += red[w][threadIdx.x];
    partials[(int64_t)threadIdx.x * nblk + blockIdx.x] = v;
  }
}

// pass B: exclusive scan of the chunk totals in chunk order, one workgroup per accumulator (blockIdx.y; the seven
// scans are independent -- one workgroup doing all of them took 0.67 ms at 5e4 chunks); total of the opening ranges
// (= sum of all ranges, registration.cc:51) -> out[0].  Same association as before: thread partials in thread
// order, then the chunks of a thread in order.
__global__ __launch_bounds__(1024) void tls_sweep_scan_kernel(double* __restrict__ partials,
                                                              int64_t nblk,
                                                              double* __restrict__ out,
                                                              const ScaleSeg* __restrict__ segs) {
  __shared__ double tot[1024];
  const int t = threadIdx.x, q = blockIdx.y;
  if (segs) {  // batch: blockIdx.x = problem slot
    const ScaleSeg sg = segs[blockIdx.x];
    partials += kNumAcc * sg.blk_off;
    nblk = sg.nblk;
    out += 2 * blockIdx.x;
  }
  const int64_t L = (nblk + 1023) / 1024;
  const int64_t b0 = (int64_t)t * L, b1 = b0 + L < nblk ? b0 + L : nblk;
  double* pq = partials + (int64_t)q * nblk;
  {
    double sum = 0;
    for (int64_t b = b0; b < b1; ++b) sum += pq[b];
    tot[t] = sum;
  }
  __syncthreads();
  if (t == 0) {
    double acc = 0;
    for (int k = 0; k < 1024; ++k) {
      const double v = tot[k];
      tot[k] = acc;
      acc += v;
    }
    if (q == 6) out[0] = acc;
  }
  __syncthreads();
  double acc = tot[t];
  for (int64_t b = b0; b < b1; ++b) {
    const double v = pq[b];
    pq[b] = acc;
    acc += v;
  }
}

struct Best {
  double cost, hat;
  int64_t pos;
};

__device__ __forceinline__ bool best_less(double ca, int64_t pa, double cb, int64_t pb) {
  return ca < cb || (ca == cb && pa < pb);  // first minimum (registration.cc:78); NaN never wins
}

// pass C: per-endpoint cost (registration.cc:58-75) and the chunk's first minimum.
__global__ __launch_bounds__(kSwThreads) void tls_sweep_cost_kernel(
    const int32_t* __restrict__ tags, const double2* __restrict__ sorted_xr,
    int64_t m, int64_t nblk, const double* __restrict__ partials,
    const double* __restrict__ ranges_sum_p, double* __restrict__ best_cost,
    double* __restrict__ best_hat, int64_t* __restrict__ best_pos, double* __restrict__ first_hat,
    const ScaleSeg* __restrict__ segs) {
  __shared__ double wtot[kSwThreads / 64][6];
  __shared__ double bc[kSwThreads / 64], bh[kSwThreads / 64];
  __shared__ int64_t bp[kSwThreads / 64];
  if (segs) {
    const ScaleSeg sg = segs[blockIdx.y];
    if ((int64_t)blockIdx.x >= sg.nblk) return;
    tags += sg.e_off;
    sorted_xr += sg.e_off;
    partials += kNumAcc * sg.blk_off;
    best_cost += sg.blk_off;
    best_hat += sg.blk_off;
    best_pos += sg.blk_off;
    ranges_sum_p += 2 * blockIdx.y;
    first_hat += 2 * blockIdx.y;
    m = sg.m;
    nblk = sg.nblk;
  }
  const int64_t base = (int64_t)blockIdx.x * kSwChunk + (int64_t)threadIdx.x * kSwPer;
  int tg[kSwPer];
  double xv[kSwPer], rv[kSwPer];
  double loc[6] = {0, 0, 0, 0, 0, 0};
  for (int e = 0; e < kSwPer; ++e) {
    tg[e] = 0;
    xv[e] = 0;
    rv[e] = 1;
    if (base + e < m) {
      const int tag = tags[base + e];
      const double2 m2 = sorted_xr[base + e];  // (sequential: gathered by pass A)
      tg[e] = tag;
      xv[e] = m2.x;
      rv[e] = m2.y;
      const double eps = tag > 0 ? 1.0 : -1.0, w = 1.0 / (rv[e] * rv[e]);
      loc[0] += eps;
      loc[1] += eps * w;
      loc[2] += eps * w * xv[e];
      loc[3] += eps * rv[e];
      loc[4] += eps * xv[e];
      loc[5] += eps * xv[e] * xv[e];
    }
  }
  // exclusive prefix of the thread totals inside the workgroup (wave scan + wave offsets)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double pre[6];
  for (int q = 0; q < 6; ++q) {
    double inc = loc[q];
    for (int off = 1; off < 64; off <<= 1) {
      const double up = shfl_up_d(inc, off);
      if (lane >= off) inc += up;
    }
    if (lane == 63) wtot[wave][q] = inc;
    pre[q] = inc - loc[q];
  }
  __syncthreads();
  const double ranges_sum = ranges_sum_p[0];
  double run[6];
  for (int q = 0; q < 6; ++q) {
    double off = partials[(int64_t)q * nblk + blockIdx.x];
    for (int w = 0; w < wave; ++w) off += wtot[w][q];
    run[q] = off + pre[q];
  }
  double bcost = INFINITY, bhat = NAN;
  int64_t bpos = INT64_MAX;
  for (int e = 0; e < kSwPer; ++e) {
    if (tg[e] == 0) continue;
    const double eps = tg[e] > 0 ? 1.0 : -1.0, w = 1.0 / (rv[e] * rv[e]);
    run[0] += eps;
    run[1] += eps * w;
    run[2] += eps * w * xv[e];
    run[3] += eps * rv[e];
    run[4] += eps * xv[e];
    run[5] += eps * xv[e] * xv[e];
    const double x_hat = run[2] / run[1];
    const double residual = run[0] * x_hat * x_hat + run[5] - 2 * run[4] * x_hat;
    const double cost = residual + (ranges_sum - run[3]);
    if (base + e == 0) first_hat[0] = x_hat;
    if (cost < bcost) {
      bcost = cost;
      bhat = x_hat;
      bpos = base + e;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double oc = shfl_down_d(bcost, off), oh = shfl_down_d(bhat, off);
    const int64_t op = __shfl_down((long long)bpos, off, 64);
    if (best_less(oc, op, bcost, bpos)) {
      bcost = oc;
      bhat = oh;
      bpos = op;
    }
  }
  if (lane == 0) {
    bc[wave] = bcost;
    bh[wave] = bhat;
    bp[wave] = bpos;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kSwThreads / 64; ++w)
      if (best_less(bc[w], bp[w], bcost, bpos)) {
        bcost = bc[w];
        bhat = bh[w];
        bpos = bp[w];
      }
    best_cost[blockIdx.x] = bcost;
    best_hat[blockIdx.x] = bhat;
    best_pos[blockIdx.x] = bpos;
  }
}

// pass D: first minimum over the chunks -> estimate
__global__ __launch_bounds__(1024) void tls_sweep_argmin_kernel(
    const double* __restrict__ best_cost, const double* __restrict__ best_hat,
    const int64_t* __restrict__ best_pos, int64_t nblk, const double* __restrict__ first_hat,
    double* __restrict__ est, const ScaleSeg* __restrict__ segs, int64_t est_stride) {
  __shared__ double bc[16], bh[16];
  __shared__ int64_t bp[16];
  if (segs) {  // batch: one workgroup per problem slot; est = the scale field of problem 0's record
    const ScaleSeg sg = segs[blockIdx.x];
    best_cost += sg.blk_off;
    best_hat += sg.blk_off;
    best_pos += sg.blk_off;
    nblk = sg.nblk;
    first_hat += 2 * blockIdx.x;
    est = reinterpret_cast<double*>(reinterpret_cast<char*>(est) + (int64_t)sg.prob * est_stride);
  }
  double c = INFINITY, h = NAN;
  int64_t p = INT64_MAX;
  for (int64_t b = threadIdx.x; b < nblk; b += 1024) {
    const double oc = best_cost[b];
    const int64_t op = best_pos[b];
    if (best_less(oc, op, c, p)) {
      c = oc;
      h = best_hat[b];
      p = op;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double oc = shfl_down_d(c, off), oh = shfl_down_d(h, off);
    const int64_t op = __shfl_down((long long)p, off, 64);
    if (best_less(oc, op, c, p)) {
      c = oc;
      h = oh;
      p = op;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    bc[wave] = c;
    bh[wave] = h;
    bp[wave] = p;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w)
      if (best_less(bc[w], bp[w], c, p)) {
        c = bc[w];
        h = bh[w];
        p = bp[w];
      }
    // no finite cost anywhere: the reference's minCoeff lands on index 0
    est[0] = (c < INFINITY) ? h : first_hat[0];
  }
}

__global__ __launch_bounds__(256) void tls_mask_kernel(const double* __restrict__ x,
                                                       const double* __restrict__ r, int64_t n,
                                                       const double* __restrict__ est,
                                                       uint8_t* __restrict__ mask) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k < n) mask[k] = fabs(x[k] - est[0]) <= r[k] ? 1 : 0;  // registration.cc:86
}

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

// Workspace layout (bytes, 256-aligned pieces): keys[2][2n] doubles, tags[2][2n] int32,
// partials [7][nblk] doubles, best cost/hat [nblk] doubles, best pos [nblk] int64, 4 scalars,
// the radix sort's temporary storage.
static size_t sort_temp_bytes(int64_t m) {
  size_t bytes = 0, fbytes = 0;
  rocprim::double_buffer<double> k(nullptr, nullptr);
  rocprim::double_buffer<float> fkb(nullptr, nullptr);
  rocprim::double_buffer<int32_t> v(nullptr, nullptr);
  (void)rocprim::radix_sort_pairs(nullptr, bytes, k, v, (size_t)m, 0, 64, (hipStream_t)0);
  (void)rocprim::radix_sort_pairs(nullptr, fbytes, fkb, v, (size_t)m, 0, 32, (hipStream_t)0);
  return bytes > fbytes ? bytes : fbytes;
}

// 64-bit sort for everything (setting scale_sort64: the A/B test of the two sort routes)
static bool force_sort64() { return setting(S_SCALE_SORT64) != 0; }

int64_t scalar_tls_large_workspace_bytes(int64_t n) {
  const int64_t m = 2 * n;
  const int64_t nblk = (m + kSwChunk - 1) / kSwChunk;
  size_t b = 0;
  b += 2 * align_up((size_t)m * 8);
  b += 2 * align_up((size_t)m * 4);
  b += align_up((size_t)m * 16);  // measurements in sorted order (float-key path)
  b += align_up((size_t)nblk * 8 * kNumAcc);
  b += 3 * align_up((size_t)nblk * 8);
  b += 256;
  b += align_up(sort_temp_bytes(m));
  return (int64_t)b;
}

namespace {
struct Work {
  double* keys[2];   // (float-key path: the first buffer holds both float key buffers)
  int32_t* tags[2];
  double2* xr_sorted;
  double* partials;
  double* best_cost;
  double* best_hat;
  int64_t* best_pos;
  double* scalars;  // [0] sum of ranges, [1] first x_hat
  void* sort_tmp;
  size_t sort_tmp_bytes;
  int64_t nblk;
};

Work carve(char* ws, int64_t n) {
  Work w;
  const int64_t m = 2 * n;
  w.nblk = (m + kSwChunk - 1) / kSwChunk;
  char* p = ws;
  for (int k = 0; k < 2; ++k) {
    w.keys[k] = reinterpret_cast<double*>(p);
    p += align_up((size_t)m * 8);
  }
  for (int k = 0; k < 2; ++k) {
    w.tags[k] = reinterpret_cast<int32_t*>(p);
    p += align_up((size_t)m * 4);
  }
  w.xr_sorted = reinterpret_cast<double2*>(p);
  p += align_up((size_t)m * 16);
  w.partials = reinterpret_cast<double*>(p);
  p += align_up((size_t)w.nblk * 8 * kNumAcc);
  w.best_cost = reinterpret_cast<double*>(p);
  p += align_up((size_t)w.nblk * 8);
  w.best_hat = reinterpret_cast<double*>(p);
  p += align_up((size_t)w.nblk * 8);
  w.best_pos = reinterpret_cast<int64_t*>(p);
  p += align_up((size_t)w.nblk * 8);
  w.scalars = reinterpret_cast<double*>(p);
  p += 256;
  w.sort_tmp = p;
  w.sort_tmp_bytes = sort_temp_bytes(m);
  return w;
}

// sort + sweep on endpoints already generated into w.keys[0] / w.tags[0] (64-bit keys), or -- float-key path,
// d_overflow != nullptr -- into fkeys(w) / w.tags[0]
inline float* fkeys0(const Work& w) { return reinterpret_cast<float*>(w.keys[0]); }
inline float* fkeys1(const Work& w, int64_t m) { return reinterpret_cast<float*>(w.keys[0]) + ((m + 63) & ~(int64_t)63); }

hipError_t sort_and_sweep(hipStream_t s, const Work& w, const double* d_x, const double* d_r,
                          int64_t n, double* d_est, int32_t* d_overflow, const double* d_src = nullptr,
                          const double* d_dst = nullptr, int n_pts = 0, double beta = 0.0) {
  const int64_t m = 2 * n;
  const int32_t* tags = nullptr;
  const double2* sorted_xr = nullptr;
  size_t tmp = w.sort_tmp_bytes;
  if (d_overflow) {
    rocprim::double_buffer<float> kb(fkeys0(w), fkeys1(w, m));
    rocprim::double_buffer<int32_t> vb(w.tags[0], w.tags[1]);
    hipError_t e = rocprim::radix_sort_pairs(w.sort_tmp, tmp, kb, vb, (size_t)m, 0, 32, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tls_order_fix_kernel, dim3((unsigned)w.nblk), dim3(kSwThreads), 0, s,
                       reinterpret_cast<const uint32_t*>(kb.current()), 1, vb.current(), d_x, d_r, m, w.nblk, vb.alternate(),
                       w.xr_sorted, d_overflow, (int64_t)0, static_cast<const ScaleSeg*>(nullptr), d_src, d_dst, n_pts, beta);
    tags = vb.alternate();
    sorted_xr = w.xr_sorted;
    hipLaunchKernelGGL(tls_sweep_totals_kernel, dim3((unsigned)w.nblk), dim3(kSwThreads), 0, s, tags,
                       static_cast<const double*>(nullptr), static_cast<const double*>(nullptr), m, w.nblk, w.partials,
                       w.xr_sorted, static_cast<const ScaleSeg*>(nullptr));
  } else {
    rocprim::double_buffer<double> kb(w.keys[0], w.keys[1]);
    rocprim::double_buffer<int32_t> vb(w.tags[0], w.tags[1]);
    hipError_t e = rocprim::radix_sort_pairs(w.sort_tmp, tmp, kb, vb, (size_t)m, 0, 64, s);
    if (e != hipSuccess) return e;
    tags = vb.current();
    // the sorted values are dead (only the order matters): the two key buffers, back to back, take the
    // measurements in sorted order (2 x 8 -> 16 bytes per endpoint)
    double2* out_xr = reinterpret_cast<double2*>(w.keys[0]);
    sorted_xr = out_xr;
    hipLaunchKernelGGL(tls_sweep_totals_kernel, dim3((unsigned)w.nblk), dim3(kSwThreads), 0, s, tags,
                       d_x, d_r, m, w.nblk, w.partials, out_xr, static_cast<const ScaleSeg*>(nullptr));
  }
  hipLaunchKernelGGL(tls_sweep_scan_kernel, dim3(1, kNumAcc), dim3(1024), 0, s, w.partials, w.nblk,
                     w.scalars, static_cast<const ScaleSeg*>(nullptr));
  hipLaunchKernelGGL(tls_sweep_cost_kernel, dim3((unsigned)w.nblk), dim3(kSwThreads), 0, s, tags,
                     sorted_xr, m, w.nblk, w.partials, w.scalars, w.best_cost, w.best_hat,
                     w.best_pos, w.scalars + 1, static_cast<const ScaleSeg*>(nullptr));
  hipLaunchKernelGGL(tls_sweep_argmin_kernel, dim3(1), dim3(1024), 0, s, w.best_cost, w.best_hat,
                     w.best_pos, w.nblk, w.scalars + 1, d_est, static_cast<const ScaleSeg*>(nullptr), (int64_t)0);
  return hipGetLastError();
}
}  // namespace

// ---- a batch of problems through ONE value sort ----------------------------------------------------------
// All the problems' endpoints are generated into one array (global tags), sorted by value in one device-wide
// stable radix sort, then gathered back into per-problem segments by a second stable pass on the problem
// slot (ceil(log2 count) bits): LSD order, so inside a segment the endpoints are sorted by value with ties in
// insertion order -- the order the one-problem path produces.  The three-pass sweep then runs on all segments
// at once (grid.y = slot), every segment cut into chunks from ITS OWN start, so that each problem's sums are
// associated exactly as on the one-problem path: the batch is bit-identical to the problems solved one by one.
void scale_batch_plan(ScaleSeg* segs, int count, int64_t* total_trims, int64_t* total_blocks, int* max_n,
                      int64_t* max_nblk) {
  int64_t trims = 0, blocks = 0, mb = 0;
  int mn = 0;
  for (int q = 0; q < count; ++q) {
    ScaleSeg& g = segs[q];
    const int64_t M = (int64_t)g.n * (g.n - 1) / 2;
    g.trim_off = trims;
    g.e_off = 2 * trims;
    g.m = 2 * M;
    g.nblk = (g.m + kSwChunk - 1) / kSwChunk;
    g.blk_off = blocks;
    trims += M;
    blocks += g.nblk;
    mn = g.n > mn ? g.n : mn;
    mb = g.nblk > mb ? g.nblk : mb;
  }
  *total_trims = trims;
  *total_blocks = blocks;
  *max_n = mn;
  *max_nblk = mb;
}

namespace {
struct BatchWork {
  double* keys[2];
  int32_t* tags[2];
  double2* xr_sorted;
  double *partials, *best_cost, *best_hat, *scalars;
  int64_t* best_pos;
  ScaleSeg* segs;
  void* sort_tmp;
  size_t sort_tmp_bytes;
};
size_t slot_sort_temp_bytes(int64_t m) {
  size_t bytes = 0;
  rocprim::double_buffer<uint32_t> k(nullptr, nullptr);
  rocprim::double_buffer<int32_t> v(nullptr, nullptr);
  (void)rocprim::radix_sort_pairs(nullptr, bytes, k, v, (size_t)m, 0, 8, (hipStream_t)0);
  return bytes;
}
BatchWork carve_batch(char* ws, int64_t trims, int64_t blocks, int count, size_t* total) {
  BatchWork w;
  const int64_t m = 2 * trims;
  char* p = ws;
  auto take = [&](size_t bytes) {
    char* r = p;
    p += align_up(bytes);
    return r;
  };
  for (int k = 0; k < 2; ++k) w.keys[k] = reinterpret_cast<double*>(take((size_t)m * 8));
  for (int k = 0; k < 2; ++k) w.tags[k] = reinterpret_cast<int32_t*>(take((size_t)m * 4));
  w.xr_sorted = reinterpret_cast<double2*>(take((size_t)m * 16));  // (float-key path)
  // (the slot keys of the second pass reuse the value keys' storage: the values are dead once sorted)
  w.partials = reinterpret_cast<double*>(take((size_t)blocks * 8 * kNumAcc));
  w.best_cost = reinterpret_cast<double*>(take((size_t)blocks * 8));
  w.best_hat = reinterpret_cast<double*>(take((size_t)blocks * 8));
  w.best_pos = reinterpret_cast<int64_t*>(take((size_t)blocks * 8));
  w.scalars = reinterpret_cast<double*>(take((size_t)count * 16));
  w.segs = reinterpret_cast<ScaleSeg*>(take((size_t)count * sizeof(ScaleSeg)));
  size_t c = 0;
  {
    rocprim::double_buffer<unsigned long long> k(nullptr, nullptr);
    rocprim::double_buffer<int32_t> v(nullptr, nullptr);
    (void)rocprim::radix_sort_pairs(nullptr, c, k, v, (size_t)m, 0, 48, (hipStream_t)0);
  }
  const size_t a = sort_temp_bytes(m), b = slot_sort_temp_bytes(m);
  w.sort_tmp_bytes = a > b ? a : b;
  if (c > w.sort_tmp_bytes) w.sort_tmp_bytes = c;
  w.sort_tmp = take(w.sort_tmp_bytes);
  *total = (size_t)(p - ws);
  return w;
}
}  // namespace

int64_t scale_batch_workspace_bytes(int64_t trims, int64_t blocks, int count) {
  size_t total = 0;
  (void)carve_batch(nullptr, trims, blocks, count, &total);
  return (int64_t)total;
}

hipError_t launch_tls_scale_batch(hipStream_t s, const double* d_src, const double* d_dst, const ScaleSeg* h_segs,
                                  int count, int64_t trims, int64_t blocks, int max_n, int64_t max_nblk, double beta,
                                  double* d_raw, double* d_alpha, char* d_workspace, double* d_scale0,
                                  int64_t scale_stride, int32_t* d_overflow0) {
  if (force_sort64()) d_overflow0 = nullptr;
  size_t total = 0;
  const BatchWork w = carve_batch(d_workspace, trims, blocks, count, &total);
  const int64_t m = 2 * trims;
  // (pageable source: the runtime stages it before returning)
  hipError_t e = hipMemcpyAsync(w.segs, h_segs, (size_t)count * sizeof(ScaleSeg), hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return e;
  int bits = 1;
  while ((1 << bits) < count) ++bits;
  size_t tmp = w.sort_tmp_bytes;
  const int32_t* tags = nullptr;
  const double2* sorted_xr = nullptr;
  if (d_overflow0) {
    // float-key path: one sort on the composite (slot, float key), then the exact order inside the float runs
    unsigned long long* ck0 = reinterpret_cast<unsigned long long*>(w.keys[0]);
    unsigned long long* ck1 = reinterpret_cast<unsigned long long*>(w.keys[1]);
    hipLaunchKernelGGL(trim_endpoints_batch_kernel, dim3((unsigned)(max_n - 1), (unsigned)count), dim3(256), 0, s, d_src,
                       d_dst, w.segs, beta, d_raw, d_alpha, w.keys[0], w.tags[0], ck0);
    rocprim::double_buffer<unsigned long long> kb(ck0, ck1);
    rocprim::double_buffer<int32_t> vb(w.tags[0], w.tags[1]);
    e = rocprim::radix_sort_pairs(w.sort_tmp, tmp, kb, vb, (size_t)m, 0, (unsigned)(32 + bits), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tls_order_fix_kernel, dim3((unsigned)max_nblk, (unsigned)count), dim3(kSwThreads), 0, s,
                       reinterpret_cast<const uint32_t*>(kb.current()), 2, vb.current(), d_raw,
                       static_cast<const double*>(nullptr), (int64_t)0, (int64_t)0, vb.alternate(), w.xr_sorted, d_overflow0,
                       scale_stride, w.segs, d_src, d_dst, 0, beta);
    tags = vb.alternate();
    sorted_xr = w.xr_sorted;
    hipLaunchKernelGGL(tls_sweep_totals_kernel, dim3((unsigned)max_nblk, (unsigned)count), dim3(kSwThreads), 0, s, tags,
                       static_cast<const double*>(nullptr), static_cast<const double*>(nullptr), (int64_t)0, (int64_t)0,
                       w.partials, w.xr_sorted, w.segs);
  } else {
    hipLaunchKernelGGL(trim_endpoints_batch_kernel, dim3((unsigned)(max_n - 1), (unsigned)count), dim3(256), 0, s, d_src,
                       d_dst, w.segs, beta, d_raw, d_alpha, w.keys[0], w.tags[0],
                       static_cast<unsigned long long*>(nullptr));
    rocprim::double_buffer<double> kb(w.keys[0], w.keys[1]);
    rocprim::double_buffer<int32_t> vb(w.tags[0], w.tags[1]);
    e = rocprim::radix_sort_pairs(w.sort_tmp, tmp, kb, vb, (size_t)m, 0, 64, s);
    if (e != hipSuccess) return e;
    uint32_t* slot_in = reinterpret_cast<uint32_t*>(kb.alternate());  // the dead half of the value keys
    uint32_t* slot_alt = reinterpret_cast<uint32_t*>(kb.current());
    hipLaunchKernelGGL(scale_slot_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, vb.current(), m, w.segs,
                       count, slot_in);
    rocprim::double_buffer<uint32_t> sb(slot_in, slot_alt);
    rocprim::double_buffer<int32_t> vb2(vb.current(), vb.alternate());
    tmp = w.sort_tmp_bytes;
    e = rocprim::radix_sort_pairs(w.sort_tmp, tmp, sb, vb2, (size_t)m, 0, (unsigned)bits, s);
    if (e != hipSuccess) return e;
    tags = vb2.current();
    double2* out_xr = reinterpret_cast<double2*>(w.keys[0]);  // (both key buffers are dead by now)
    sorted_xr = out_xr;
    hipLaunchKernelGGL(tls_sweep_totals_kernel, dim3((unsigned)max_nblk, (unsigned)count), dim3(kSwThreads), 0, s, tags,
                       d_raw, static_cast<const double*>(nullptr), (int64_t)0, (int64_t)0, w.partials, out_xr, w.segs);
  }
  hipLaunchKernelGGL(tls_sweep_scan_kernel, dim3((unsigned)count, kNumAcc), dim3(1024), 0, s, w.partials, (int64_t)0, w.scalars,
                     w.segs);
  hipLaunchKernelGGL(tls_sweep_cost_kernel, dim3((unsigned)max_nblk, (unsigned)count), dim3(kSwThreads), 0, s, tags,
                     sorted_xr, (int64_t)0, (int64_t)0, w.partials, w.scalars, w.best_cost, w.best_hat, w.best_pos,
                     w.scalars + 1, w.segs);
  hipLaunchKernelGGL(tls_sweep_argmin_kernel, dim3((unsigned)count), dim3(1024), 0, s, w.best_cost, w.best_hat,
                     w.best_pos, (int64_t)0, w.scalars + 1, d_scale0, w.segs, scale_stride);
  return hipGetLastError();
}

// Scalar TLS of n measurements x[n] with ranges r[n] (device arrays) -> d_est (+ optional mask).
// d_overflow: nullptr = the 64-bit sort; else the float-key path, *d_overflow (zeroed by the caller) is set when a
// run of equal float keys was too long to fix (the caller repeats the call with nullptr).
hipError_t launch_scalar_tls_large(hipStream_t s, const double* d_x, const double* d_r, int64_t n,
                                   char* d_workspace, double* d_est, uint8_t* d_mask, int32_t* d_overflow) {
  if (force_sort64()) d_overflow = nullptr;
  const Work w = carve(d_workspace, n);
  hipLaunchKernelGGL(tls_endpoints_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_x,
                     d_r, n, w.keys[0], w.tags[0], d_overflow ? fkeys0(w) : static_cast<float*>(nullptr));
  hipError_t e = sort_and_sweep(s, w, d_x, d_r, n, d_est, d_overflow);
  if (e != hipSuccess) return e;
  if (d_mask)
    hipLaunchKernelGGL(tls_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_x, d_r,
                       n, d_est, d_mask);
  return hipGetLastError();
}

// TLS scale estimate of one problem's n points: TRIMs + endpoints fused, then sort + sweep.
// d_raw: [M] (raw, alpha) pairs = 16 M bytes (the sweep gathers from it once); d_alpha: unused.
hipError_t launch_tls_scale_large(hipStream_t s, const double* d_src, const double* d_dst, int n,
                                  double beta, double* d_raw, double* d_alpha, char* d_workspace,
                                  double* d_scale, int32_t* d_overflow) {
  if (force_sort64()) d_overflow = nullptr;
  const int64_t M = (int64_t)n * (n - 1) / 2;
  const Work w = carve(d_workspace, M);
  hipLaunchKernelGGL(trim_endpoints_kernel, dim3(n - 1), dim3(256), 0, s, d_src, d_dst, n, beta,
                     d_raw, d_alpha, w.keys[0], w.tags[0], d_overflow ? fkeys0(w) : static_cast<float*>(nullptr));
  // (64-bit path: (raw, alpha) interleaved in d_raw; float-key path: recomputed from the points)
  return sort_and_sweep(s, w, d_raw, nullptr, M, d_scale, d_overflow, d_src, d_dst, n, beta);
}

// Stage entry point solveForScale(v1, v2) on caller-supplied TIMs (registration.h:584, registration.cc
// 410-443): per TIM k the two norms, then either the TRIM terms of TLSScaleSolver (raw = |v2|/|v1|,
// alpha = beta * (1/|v1|), :415-422) or the mask of ScaleInliersSelector (| |v1| - |v2| | <= beta, :442).
// Norms as the reference's colwise sums: (x^2 + y^2) + z^2, individually rounded, IEEE sqrt.
__global__ __launch_bounds__(256) void tim_scale_terms_kernel(const double* __restrict__ v1,
                                                              const double* __restrict__ v2, int64_t m,
                                                              double beta, int estimate,
                                                              double* __restrict__ raw,
                                                              double* __restrict__ alpha,
                                                              uint8_t* __restrict__ mask) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= m) return;
  const double ax = v1[3 * k], ay = v1[3 * k + 1], az = v1[3 * k + 2];
  const double bx = v2[3 * k], by = v2[3 * k + 1], bz = v2[3 * k + 2];
  const double n1 = __builtin_sqrt((ax * ax + ay * ay) + az * az);
  const double n2 = __builtin_sqrt((bx * bx + by * by) + bz * bz);
  if (estimate) {
    raw[k] = n2 / n1;
    alpha[k] = beta * (1.0 / n1);
  } else {
    mask[k] = __builtin_fabs(n1 - n2) <= beta ? 1 : 0;
  }
}

void launch_tim_scale_terms(hipStream_t s, const double* d_v1, const double* d_v2, int64_t m, double beta,
                            int estimate, double* d_raw, double* d_alpha, uint8_t* d_mask) {
  if (m <= 0) return;
  hipLaunchKernelGGL(tim_scale_terms_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, d_v1, d_v2, m,
                     beta, estimate, d_raw, d_alpha, d_mask);
}

}  // namespace thip
