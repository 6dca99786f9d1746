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
#include <algorithm>
#include <cstdio>
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
// BY_ID (the hull path: the endpoints arrive compacted in arbitrary order, and the arrays are padded with tag-0 entries up
// to a fixed capacity): equal FP64 keys are ranked by endpoint id 2 k + (closing ? 1 : 0) -- the insertion order of the
// reference's loop -- instead of by position.
template <bool BY_ID>
__global__ __launch_bounds__(kSwThreads) void tls_order_fix_kernel(
    const uint32_t* __restrict__ fk, int fk_stride, const int32_t* __restrict__ tags, const double* __restrict__ x,
    const double* __restrict__ r, int64_t m, int64_t nblk, int32_t* __restrict__ out_tags,
    double2* __restrict__ out_xr, int32_t* __restrict__ overflow, int64_t overflow_stride,
    const ScaleSeg* __restrict__ segs, const double* __restrict__ src, const double* __restrict__ dst, int n_pts,
    double beta) {
  __shared__ uint32_t s_fk[kSwChunk + 2 * kFxHalo];
  __shared__ double s_kd[kSwChunk + 2 * kFxHalo];
  __shared__ uint32_t s_id[BY_ID ? kSwChunk + 2 * kFxHalo : 1];
  auto id_of = [](int tag) -> uint32_t { return tag > 0 ? 2u * (uint32_t)(tag - 1) : 2u * (uint32_t)(-tag - 1) + 1u; };
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
      if (BY_ID) s_id[li] = id_of(tg[e]);
    } else if (BY_ID && pos < m) {
      s_fk[li] = 0xffffffffu;  // padding: no key of a real endpoint (the pad keys are 0x7f7f7f7f)
    }
  }
  if (threadIdx.x < 2 * kFxHalo) {  // the halos
    const int li = threadIdx.x < kFxHalo ? threadIdx.x : kSwChunk + threadIdx.x;
    const int64_t pos = lo + li;
    if (pos >= 0 && pos < m) {
      const int t = tags[pos];
      if (!BY_ID || t != 0) {
        const double2 v = measure(t);
        s_fk[li] = fk[pos * fk_stride];
        s_kd[li] = t > 0 ? v.x - v.y : v.x + v.y;
        if (BY_ID) s_id[li] = id_of(t);
      } else {
        s_fk[li] = 0xffffffffu;
      }
    }
  }
  __syncthreads();
  bool over = false;
  for (int e = 0; e < kSwPer; ++e) {
    if (tg[e] == 0) {
      if (BY_ID) {  // padding stays padding (the sweep skips tag 0)
        const int64_t ppos = c0 + threadIdx.x + (int64_t)e * kSwThreads;
        if (ppos < m) out_tags[ppos] = 0;
      }
      continue;
    }
    const int li = kFxHalo + threadIdx.x + e * kSwThreads;
    const int64_t pos = lo + li;
    const uint32_t f = s_fk[li];
    const double kd = s_kd[li];
    int shift = 0;
    // items of the run in front of this one that must end up behind it (strictly larger key) ...
    int j = li - 1;
    const uint32_t my_id = BY_ID ? s_id[li] : 0u;
    for (; j >= 0 && lo + j >= 0 && s_fk[j] == f; --j)
      shift -= (s_kd[j] > kd || (BY_ID && s_kd[j] == kd && s_id[j] > my_id)) ? 1 : 0;
    if (j < 0 && lo + j >= 0) over = true;  // the run continues beyond the halo
    // ... and behind it that must end up in front (strictly smaller key)
    j = li + 1;
    for (; j < kSwChunk + 2 * kFxHalo && lo + j < m && s_fk[j] == f; ++j)
      shift += (s_kd[j] < kd || (BY_ID && s_kd[j] == kd && s_id[j] < my_id)) ? 1 : 0;
    if (j >= kSwChunk + 2 * kFxHalo && lo + j < m) over = true;
    out_tags[pos + shift] = tg[e];
    out_xr[pos + shift] = xr[e];
  }
  if (over) *overflow = 1;
}

// ------------------------------------------------------------------------------------------
// Hull of the arg-min (setting scale_hull, large single problems): the sweep of registration.cc:58-78 needs the
// endpoints in exact order only where the minimum can be.  cost(e) = SS(I_e) + R_total - S(e), S(e) = the summed ranges
// of the consensus set I_e, SS its sum of squares about x_hat, so inside a value bin b
//     cost >= R_total - (S at the bin's start + the ranges of the openers inside b) + SS_mean(members at the start that
//             do not close inside b)
// (the consensus set only grows by the bin's openers; the sum of squares about the mean never falls when members are
// added, and it is below the sum about any other point).  Bins whose bound exceeds ONE achieved cost cannot hold the
// minimum; the rest, first to last, is the hull [t_lo, t_hi): only its endpoints are sorted, order-fixed and swept, from
// the exact state in front of t_lo.  Four passes regenerate the TRIMs from the points (nothing is stored per TRIM):
//   hull_hist_kernel   per bin, openers and closers apart: ranges (openers rounded up, closers down), counts, and --
//                      for the measurements within kHullSpan of the centre -- count, sum, sum of squares; 64-bit
//                      FIXED-POINT integers in LDS, merged by integer atomics: sums that do not depend on any order
//   hull_eval_kernel   the achieved cost: the consensus set in front of the boundary of the largest S, summed in FP64 in
//                      a fixed order (rows, then a fixed tree)
//   hull_plan_kernel   bounds, hull, its size against the capacity of the compacted arrays
//   hull_emit_kernel   the hull's endpoints (float key, tag) compacted in arbitrary order -- the order-fix pass then ranks
//                      equal keys by ENDPOINT ID, i.e. insertion order, instead of by position -- and, per row, the state
//                      contribution of everything in front of the hull
// Anything unusual (a range beyond kHullRangeCap, a non-finite measurement, a hull larger than the capacity, no finite
// cost) raises the problem's overflow flag and the host repeats the stage on the full 64-bit sort, like a float-key run
// too long to fix.  Sized before it was built: scripts/probe/scale_hull_model.py (profiles/r6b/scale_hull_model.txt).
// ------------------------------------------------------------------------------------------
constexpr int kHullBins = 2048;
constexpr int kHullTail = 256;               // geometric bins above 4 c (c = ratio of the clouds' RMS sizes)
constexpr double kHullRangeCap = 1024.0;     // ranges are accumulated with 20 fractional bits
constexpr double kHullSpan = 64.0;           // |x - c| <= kHullSpan c takes part in the sums of squares
constexpr int kHullThreads = 1024;
static_assert(kHullBins == 2048 && kHullThreads == 1024, "hull_plan_kernel's prefix sums: 4 kinds x 256 threads x 8 bins");
struct HullPlan {
  double c;            // centre of the measurements (ratio of the RMS sizes of dst and src)
  double v0, inv_w;    // the table's linear part: T[b] ~ v0 + b / inv_w for 1 <= b <= n_lin
  double t0;           // boundary whose consensus set gives the achieved cost
  double ub;           // that cost
  double r_total;      // sum of all ranges
  double t_lo, t_hi;   // the hull
  long long hull_items;
  int first_bin, last_bin;
  int anomalies;       // measurements the histograms cannot take
  int failed;          // 1: no hull (the caller's overflow flag is raised as well)
  unsigned int emitted;  // (filled by hull_init_kernel: the sum of the shard counters)
  int n_lin;
  int levels, pad;     // 1 / 2: which level's hull t_lo, t_hi describe
};
// The compacted arrays are cut into kHullShards equal shards, row i appends to shard i mod kHullShards: one atomic
// per wave and 256 rows of pairs on ONE counter serialised in L2 (7.8e5 returning atomics: 8.9 ms for a pass whose
// arithmetic takes 0.2); a shard that fills up fails the hull (the rows are spread evenly: it takes a hull within a few
// per cent of the capacity).  Unused slots keep their padding (tag 0), which the sort moves behind every real key.
constexpr int kHullShards = 256;  // (a power of two)
constexpr int kHullEmitSplit = 4;
constexpr int kHullCountStride = 64;  // uint32 per shard counter: one 256-byte line each (neighbours in one line serialise in L2)
// histograms, [kind][bin]: 0 / 1 ranges of openers (rounded up) / closers (rounded down); 2 / 3 counts, 4 / 5 sums, 6 / 7
// sums of squares of the openers / closers within the span (8 x 16 KB + the 16 KB table: one workgroup per CU)
constexpr int kHullKinds = 8;

// largest b with T[b] <= v (T[0] = -inf, T[kHullBins] = +inf; NaN ends in the last bin).  The table IS the definition of the
// bins (every kernel asks it, so they agree to the bit); `inv_w` = bins per unit of the table's linear part, `v0` its
// origin, give a first guess that is off by at most a step there, the geometric tail is searched
__device__ __forceinline__ int hull_bin(const double* __restrict__ T, double v, double v0, double inv_w, int n_lin) {
  if (!(v == v)) return kHullBins - 1;
  int lo = 0, hi = kHullBins;  // invariant: T[lo] <= v < T[hi]
  const double g = (v - v0) * inv_w;
  if (g >= 1.0 && g < (double)n_lin) {
    int b = (int)g;
    b = b < 1 ? 1 : (b > n_lin ? n_lin : b);
    if (T[b] <= v) {
      if (v < T[b + 1]) return b;
      if (v < T[b + 2]) return b + 1;
      lo = b + 1;
    } else {
      if (T[b - 1] <= v) return b - 1;
      hi = b - 1;
    }
  }
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (T[mid] <= v) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void hull_table_kernel(const double* __restrict__ src, const double* __restrict__ dst, int n,
                                                         double* __restrict__ T, HullPlan* __restrict__ plan) {
  __shared__ double red[4][8];
  // centre: sqrt(sum |d - mean d|^2 / sum |s - mean s|^2) from one-pass sums (only a location for the bins: any positive
  // value gives a correct result)
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int p = threadIdx.x; p < n; p += 256) {
    for (int k = 0; k < 3; ++k) {
      const double sv = src[3 * p + k], dv = dst[3 * p + k];
      a[k] += sv;
      a[3 + k] += dv;
      a[6] += sv * sv;
      a[7] += dv * dv;
    }
  }
  for (int q = 0; q < 8; ++q) {
    double v = a[q];
    for (int off = 32; off > 0; off >>= 1) v += shfl_down_d(v, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = v;
  }
  __syncthreads();
  __shared__ double c_sh;
  if (threadIdx.x == 0) {
    double t[8];
    for (int q = 0; q < 8; ++q) t[q] = red[0][q] + red[1][q] + red[2][q] + red[3][q];
    const double vs = t[6] - (t[0] * t[0] + t[1] * t[1] + t[2] * t[2]) / n;
    const double vd = t[7] - (t[3] * t[3] + t[4] * t[4] + t[5] * t[5]) / n;
    double c = __builtin_sqrt(vd / vs);
    if (!(c > 1e-300 && c < 1e300)) c = 1.0;
    c_sh = c;
    plan->c = c;
    plan->v0 = 0.0;
    plan->inv_w = (double)(kHullBins - kHullTail) / (4.0 * c);
    plan->n_lin = kHullBins - kHullTail;
    plan->anomalies = 0;
    plan->failed = 0;
    plan->emitted = 0u;
    plan->hull_items = 0;
  }
  __syncthreads();
  const double c = c_sh;
  constexpr int kLin = kHullBins - kHullTail;
  for (int b = threadIdx.x; b <= kHullBins; b += 256) {
    double v;
    if (b == 0) v = -INFINITY;
    else if (b == kHullBins) v = INFINITY;
    else if (b <= kLin) v = (4.0 * c) * ((double)b / kLin);
    else v = (4.0 * c) * exp(log(2500.0) * ((double)(b - kLin) / kHullTail));
    T[b] = v;
  }
}

// round-to-nearest-even of |x| < 2^51 as an integer: one addition and one integer subtraction (the compiler's own
// FP64 -> int64 conversion is a ~20-instruction sequence, and the histogram pass converts five values per pair)
__device__ __forceinline__ long long rint_small_ll(double x, double* rounded) {
  const double y = x + 6755399441055744.0;  // 2^52 + 2^51
  *rounded = y - 6755399441055744.0;
  return __double_as_longlong(y) - 0x4338000000000000ll;
}

// INNER = level 2: bins 0 and kHullBins - 1 (outside the first hull: nine endpoints in ten) are left out.
// What a pass costs beyond the measurements themselves (hull_eval_kernel makes the same ones alone in 0.205 ms = the FP64
// issue rate) is the binning code -- table searches, conversions, eight 64-bit LDS atomics -- and a wave runs it whenever
// ONE of its lanes needs it.  Level 2 therefore COMPACTS: the pairs with an endpoint inside the hull (two compares
// against the hull's ends: exactly hull_bin's own criterion for the two outer bins; a NaN fails both and is kept) are
// queued in the wave's 1 KB of LDS and binned up to 63 at a time (0.76 -> 0.48 ms; level 1: 0.65 -> 0.61 with the cheaper conversions).  Measured and not adopted (profiles/r6c): four
// pairs per thread side by side (not latency), per-workgroup slabs instead of global atomics for the flush (not contention).
constexpr int kHullQueue = 63;
template <bool INNER>
__global__ __launch_bounds__(kHullThreads) void hull_hist_kernel(const double* __restrict__ src, const double* __restrict__ dst,
                                                                 int n, double beta, const double* __restrict__ T_g,
                                                                 HullPlan* __restrict__ plan,
                                                                 unsigned long long* __restrict__ hist /* [kHullKinds][kHullBins] */) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long lh[];  // [kHullKinds][kHullBins], the table, the waves' queues
  double* T = reinterpret_cast<double*>(lh + kHullKinds * kHullBins);
  // (INNER only) kHullQueue pairs per wave: what the 160 KB leave (a step in which all 64 lanes need the code bins them directly)
  double2* queue = reinterpret_cast<double2*>(T + kHullBins + 2) + (threadIdx.x >> 6) * kHullQueue;
  for (int k = threadIdx.x; k < kHullKinds * kHullBins; k += kHullThreads) lh[k] = 0ull;
  for (int k = threadIdx.x; k <= kHullBins; k += kHullThreads) T[k] = T_g[k];
  __syncthreads();
  const double c = plan->c, v0 = plan->v0, inv_w = plan->inv_w;
  const double inv_c = 1.0 / c;
  const int n_lin = plan->n_lin;
  const double t_in_lo = T[1], t_in_hi = T[kHullBins - 1];
  const int lane = threadIdx.x & 63;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  int bad = 0;
  auto bin_pair = [&](double sv, double av) {
    const double lo = sv - av, hi = sv + av;  // (the endpoint kernels' keys)
    const int bo = hull_bin(T, lo, v0, inv_w, n_lin), bc = hull_bin(T, hi, v0, inv_w, n_lin);
    const double rq = av * 1048576.0;  // 2^20; 0 <= rq <= 2^30
    double rr;
    const long long ri = rint_small_ll(rq, &rr);
    const unsigned long long r_up = (unsigned long long)(ri + (rr < rq ? 1 : 0)), r_dn = (unsigned long long)(ri - (rr > rq ? 1 : 0));  // ceil, floor
    const bool use_o = !INNER || (bo >= 1 && bo <= kHullBins - 2), use_c = !INNER || (bc >= 1 && bc <= kHullBins - 2);
    if (use_o) atomicAdd(&lh[0 * kHullBins + bo], r_up);
    if (use_c) atomicAdd(&lh[1 * kHullBins + bc], r_dn);
    const double xc = sv - c;
    if (__builtin_fabs(xc) <= kHullSpan * c) {
      // centred, scaled by c: |xs| <= 64 (to an ulp: the product with 1 / c instead of a division moves a quantised value
      // by one step in 2^40 at most once in ~1e4 pairs, far inside the bounds' own rounding allowance); 2^40 and 2^30
      // steps: sums of 5e7 terms stay below 2^63
      const double xs = xc * inv_c;
      double unused;
      const long long xq = rint_small_ll(xs * 1099511627776.0, &unused);         // 2^40
      const long long xxq = rint_small_ll(xs * xs * 1073741824.0, &unused);      // 2^30
      if (use_o) {
        atomicAdd(&lh[2 * kHullBins + bo], 1ull);
        atomicAdd(&lh[4 * kHullBins + bo], (unsigned long long)xq);
        atomicAdd(&lh[6 * kHullBins + bo], (unsigned long long)xxq);
      }
      if (use_c) {
        atomicAdd(&lh[3 * kHullBins + bc], 1ull);
        atomicAdd(&lh[5 * kHullBins + bc], (unsigned long long)xq);
        atomicAdd(&lh[7 * kHullBins + bc], (unsigned long long)xxq);
      }
    }
  };
  int queued = 0;  // (wave-uniform)
  auto flush = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // wave-private queue: same-wave ordering suffices
    if (lane < queued) {
      const double2 e = queue[lane];
      bin_pair(e.x, e.y);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    queued = 0;
  };
  for (int i = blockIdx.x; i < n - 1; i += gridDim.x) {
    for (int j0 = i + 1; j0 < n; j0 += kHullThreads) {  // (a wave steps through the row together: its queue is uniform)
      const int j = j0 + (int)threadIdx.x;
      bool need = false;
      double sv = 0.0, av = 0.0;
      if (j < n) {
        trim_terms(src, dst, i, j, beta, &sv, &av);
        if (!(av >= 0.0 && av <= kHullRangeCap) || !(sv == sv) || !(sv < INFINITY && sv > -INFINITY)) {
          ++bad;
        } else if (!INNER) {
          bin_pair(sv, av);
        } else {
          const double lo = sv - av, hi = sv + av;
          need = !((lo < t_in_lo || lo >= t_in_hi) && (hi < t_in_lo || hi >= t_in_hi));
        }
      }
      if (INNER) {
        const uint64_t m = __ballot(need);
        const int cnt = __popcll(m);
        if (queued + cnt > kHullQueue) flush();
        if (cnt > kHullQueue) {
          bin_pair(sv, av);  // (all 64 lanes)
        } else {
          if (need) queue[queued + __popcll(m & lt_mask)] = make_double2(sv, av);
          queued += cnt;
        }
      }
    }
  }
  if (INNER) flush();
  __syncthreads();
  for (int k = threadIdx.x; k < kHullKinds * kHullBins; k += kHullThreads) {
    const unsigned long long v = lh[k];
    if (v) atomicAdd(&hist[k], v);
  }
  if (bad && !INNER) atomicAdd(&plan->anomalies, bad);
}

// Partial state sums of one row of pairs, reduced over the workgroup in a fixed shape (thread order inside a wave by a
// shuffle tree, then the waves in order): kind = which endpoints take part
//   hull_eval_kernel: the consensus set in front of t0 (opened before t0, not closed before t0), + the sum of all ranges
//   hull_emit_kernel: every endpoint in front of t_lo, with its sign
template <int kThreads>
__device__ __forceinline__ void hull_block_reduce(double (&a)[7], double (*red)[7], double* out) {
  for (int q = 0; q < 7; ++q) {
    double v = a[q];
    for (int off = 32; off > 0; off >>= 1) v += shfl_down_d(v, off);
    a[q] = v;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
    for (int q = 0; q < 7; ++q) red[wave][q] = a[q];
  __syncthreads();
  if (threadIdx.x < 7) {
    double v = 0;
    for (int w = 0; w < kThreads / 64; ++w) v += red[w][threadIdx.x];
    out[threadIdx.x] = v;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void hull_eval_kernel(const double* __restrict__ src, const double* __restrict__ dst, int n,
                                                        double beta, const HullPlan* __restrict__ plan,
                                                        double* __restrict__ rows /* [n - 1][7] */) {
  __shared__ double red[4][7];
  const int i = blockIdx.x;
  const double t0 = plan->t0;
  double a[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int j = i + 1 + threadIdx.x; j < n; j += 256) {
    double sv, av;
    trim_terms(src, dst, i, j, beta, &sv, &av);
    a[6] += av;
    if (sv - av < t0 && !(sv + av < t0)) {
      const double w = 1.0 / (av * av);
      a[0] += 1.0;
      a[1] += w;
      a[2] += w * sv;
      a[3] += av;
      a[4] += sv;
      a[5] += sv * sv;
    }
  }
  hull_block_reduce<256>(a, red, rows + (size_t)i * 7);
}

// rows[r][7] summed over r in a fixed order (the first 256 threads: contiguous row ranges in thread order, then the threads in order)
constexpr int kHullRowThreads = 256;
__device__ __forceinline__ void hull_rows_total(const double* __restrict__ rows, int nrows, double (*tot)[7], double* out7) {
  const int t = threadIdx.x;
  const int L = (nrows + kHullRowThreads - 1) / kHullRowThreads;
  if (t < kHullRowThreads) {
    double a[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int r = t * L; r < nrows && r < (t + 1) * L; ++r)
      for (int q = 0; q < 7; ++q) a[q] += rows[(size_t)r * 7 + q];
    for (int q = 0; q < 7; ++q) tot[t][q] = a[q];
  }
  __syncthreads();
  if (t < 7) {
    double v = 0;
    for (int k = 0; k < kHullRowThreads; ++k) v += tot[k][t];
    out7[t] = v;
  }
  __syncthreads();
}

// phase 0: the four prefix sums, the boundary in front of which the summed ranges of the consensus set are largest (-> t0);
// phase 1: the achieved cost, the bounds, the hull;
// phase 2 (second level: `hist` holds the endpoints INSIDE the first hull, binned by a table that cuts it into
//          kHullBins - 2 equal bins -- bin 0 is everything in front of it, the last bin everything behind): prefix sums
//          continued from the first level's state in front of the hull (pre[.][first_bin]: exact integers), the same
//          bounds against the same achieved cost, a hull inside the hull
__global__ __launch_bounds__(kHullThreads) void hull_plan_kernel(int phase, const unsigned long long* __restrict__ hist,
                                                                 long long* __restrict__ pre /* [4][kHullBins] */,
                                                                 const double* __restrict__ T, const double* __restrict__ rows,
                                                                 int nrows, HullPlan* __restrict__ plan,
                                                                 int32_t* __restrict__ overflow) {
  // pre: exclusive prefix of (openers - closers) per bin: ranges, and -- within the span -- count, sum, sum of squares
  __shared__ double tot[kHullRowThreads][7];
  __shared__ double acc7[7];
  __shared__ int cand_first, cand_last;
  __shared__ unsigned long long red64[kHullThreads / 64];
  __shared__ long long part[4][256];
  __shared__ long long base4[4];
  const int t = threadIdx.x;
  if (plan->anomalies != 0 || plan->failed) {
    if (t == 0) {
      plan->failed = 1;
      *overflow = 1;
    }
    return;
  }
  if (phase == 0 || phase == 2) {
    // four exclusive prefix sums over the bins (integers: any order gives the same sums): 256 threads per kind, 8 bins each
    if (t < 4) base4[t] = phase == 2 ? pre[t * kHullBins + plan->first_bin] : 0;
    __syncthreads();
    const int kind = t >> 8, u = t & 255, ko = 2 * kind, kc = ko + 1;
    long long v[8], sum = 0;
    for (int e = 0; e < 8; ++e) {
      const int b = u * 8 + e;
      v[e] = (long long)hist[ko * kHullBins + b] - (long long)hist[kc * kHullBins + b];
      sum += v[e];
    }
    part[kind][u] = sum;
    __syncthreads();
    if (u == 0) {
      long long acc = base4[kind];
      for (int k = 0; k < 256; ++k) {
        const long long x = part[kind][k];
        part[kind][k] = acc;
        acc += x;
      }
    }
    __syncthreads();
    long long acc = part[kind][u];
    for (int e = 0; e < 8; ++e) {
      pre[kind * kHullBins + u * 8 + e] = acc;
      acc += v[e];
    }
    __syncthreads();
  }
  if (phase == 0) {
    // first bin with the largest start value
    unsigned long long best = 0;
    for (int b = t; b < kHullBins; b += kHullThreads) {
      const long long v = pre[b] < 0 ? 0 : pre[b];
      const unsigned long long key = ((unsigned long long)v << 11) | (unsigned long long)(kHullBins - 1 - b);
      best = key > best ? key : best;
    }
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long o = __shfl_down(best, off, 64);
      best = o > best ? o : best;
    }
    if ((t & 63) == 0) red64[t >> 6] = best;
    __syncthreads();
    if (t == 0) {
      for (int w = 1; w < kHullThreads / 64; ++w) best = red64[w] > best ? red64[w] : best;
      int b0 = kHullBins - 1 - (int)(best & 2047ull);
      if (b0 < 1) b0 = 1;  // (T[0] = -inf: nothing in front of it)
      plan->t0 = T[b0];
    }
    return;
  }
  if (phase == 1) {
    hull_rows_total(rows, nrows, tot, acc7);
    if (t == 0) {
      const double cnt = acc7[0], w = acc7[1], wx = acc7[2], rin = acc7[3], sx = acc7[4], sxx = acc7[5], rtot = acc7[6];
      const double xh = wx / w;
      const double ub = (cnt * xh * xh + sxx - 2 * sx * xh) + (rtot - rin);  // registration.cc:69-72
      plan->ub = (cnt > 0.0) ? ub : INFINITY;
      plan->r_total = rtot;
    }
  }
  if (t == 0) {
    cand_first = kHullBins;
    cand_last = -1;
  }
  __syncthreads();
  const double ub = plan->ub, rtot = plan->r_total, c = plan->c;
  if (!(ub < INFINITY) || !(ub == ub)) {
    if (t == 0) {
      plan->failed = 1;
      *overflow = 1;
    }
    return;
  }
  // every sum below errs on the safe side: ranges of openers were rounded up and of closers down, the squares carry
  // their worst-case rounding, and the comparison a relative margin far above FP64's own noise in the sweep
  const double margin = 1e-9 * __builtin_fabs(ub) + 1e-9 * rtot + 1e-6;
  const int b_lo = phase == 2 ? 1 : 0, b_hi = phase == 2 ? kHullBins - 2 : kHullBins - 1;  // (level 2: the interior bins)
  for (int b = b_lo + t; b <= b_hi; b += kHullThreads) {
    const double s_hat = ((double)pre[b] + (double)hist[0 * kHullBins + b]) * (1.0 / 1048576.0);
    const long long nc = pre[1 * kHullBins + b] - (long long)hist[3 * kHullBins + b];
    double ss = 0.0;
    if (nc > 1) {
      const double s1 = (double)(pre[2 * kHullBins + b] - (long long)hist[5 * kHullBins + b]) * (1.0 / 1099511627776.0);
      const double s2 = (double)(pre[3 * kHullBins + b] - (long long)hist[7 * kHullBins + b]) * (1.0 / 1073741824.0);
      const double err = (double)nc * (1.0 / 1073741824.0) + 2.0 * __builtin_fabs(s1) * (1.0 / 1099511627776.0) + 1e-9 * s2;
      ss = (s2 - s1 * s1 / (double)nc - err) * (c * c);  // (the sums are in units of c)
      if (!(ss > 0.0)) ss = 0.0;
    }
    const double lb = rtot - s_hat + ss;
    if (lb <= ub + margin) {
      atomicMin(&cand_first, b);
      atomicMax(&cand_last, b);
    }
  }
  __syncthreads();
  if (t == 0) {
    const int f = cand_first, l = cand_last;
    if (l < f) {  // cannot be (the bin in front of t0 is a candidate); be safe
      plan->failed = 1;
      *overflow = 1;
      return;
    }
    plan->first_bin = f;
    plan->last_bin = l;
    plan->t_lo = T[f];
    plan->t_hi = T[l + 1];
    // how many endpoints the hull holds, to within the few whose measurement lies outside the span (the host sizes the
    // compacted arrays from it, with slack; the shards' own check is what guards them)
    long long est = 0;
    for (int b = f; b <= l; ++b) est += (long long)hist[2 * kHullBins + b] + (long long)hist[3 * kHullBins + b];
    plan->hull_items = est;
    plan->levels = phase == 2 ? 2 : 1;
  }
}

// second-level table: bin 0 = (-inf, t_lo), bins 1 .. kHullBins - 2 cut [t_lo, t_hi) into equal parts, the last bin = [t_hi, inf)
__global__ __launch_bounds__(256) void hull_table2_kernel(double* __restrict__ T, HullPlan* __restrict__ plan) {
  const double lo = plan->t_lo, hi = plan->t_hi;
  constexpr int kIn = kHullBins - 2;
  const double w = (hi - lo) / kIn;
  for (int b = threadIdx.x; b <= kHullBins; b += 256) {
    double v;
    if (b == 0) v = -INFINITY;
    else if (b == kHullBins) v = INFINITY;
    else if (b == kHullBins - 1) v = hi;
    else v = lo + w * (double)(b - 1);
    if (b >= 1 && b < kHullBins - 1 && !(v < hi)) v = hi;  // (rounding: never beyond the hull's end)
    T[b] = v;
  }
  if (threadIdx.x == 0) {
    plan->v0 = lo - w;
    plan->inv_w = 1.0 / w;
    plan->n_lin = kIn;
  }
}

__global__ __launch_bounds__(256) void hull_emit_kernel(const double* __restrict__ src, const double* __restrict__ dst, int n,
                                                        double beta, HullPlan* __restrict__ plan, float* __restrict__ fkeys,
                                                        int32_t* __restrict__ tags, double* __restrict__ rows /* [n - 1][7] */,
                                                        long long cap_items, unsigned int* __restrict__ shard_count) {
  __shared__ double red[4][7];
  // kHullEmitSplit workgroups per row, each taking every kHullEmitSplit-th chunk of 256 pairs: a workgroup's reservations
  // (one returning atomic per wave and chunk) form a dependent chain, 40 links long for a whole row at N = 10 000
  const int i = blockIdx.x;
  const long long shard_items = cap_items / kHullShards;
  // rows get shorter with i: shards are assigned in snake order (0 .. 255, 255 .. 0, ...) so that every shard takes
  // the same share of long and short rows (i mod 256 alone leaves the first shard 5 % fuller than the last)
  const int shard = ((i / kHullShards) & 1) ? kHullShards - 1 - (i % kHullShards) : (i % kHullShards);
  const long long shard_base = (long long)shard * shard_items;
  unsigned int* counter = shard_count + (size_t)shard * kHullCountStride;
  const bool failed = plan->failed != 0;
  const double t_lo = plan->t_lo, t_hi = plan->t_hi;
  const int64_t seg = (int64_t)i * n - (int64_t)i * (i + 1) / 2;
  double a[7] = {0, 0, 0, 0, 0, 0, 0};
  const int lane = threadIdx.x & 63;
  for (int j0 = i + 1 + 256 * (int)blockIdx.y; j0 < n && !failed; j0 += 256 * kHullEmitSplit) {  // (block-uniform trip count)
    const int j = j0 + threadIdx.x;
    double sv = 0, av = 1, lo = 0, hi = 0;
    bool in_lo = false, in_hi = false;
    if (j < n) {
      trim_terms(src, dst, i, j, beta, &sv, &av);
      lo = sv - av;
      hi = sv + av;
      in_lo = lo >= t_lo && lo < t_hi;
      in_hi = hi >= t_lo && hi < t_hi;
      const double w = 1.0 / (av * av);
      if (lo < t_lo) {  // an opener in front of the hull
        a[0] += 1.0; a[1] += w; a[2] += w * sv; a[3] += av; a[4] += sv; a[5] += sv * sv;
      }
      if (hi < t_lo) {  // ... and a closer
        a[0] -= 1.0; a[1] -= w; a[2] -= w * sv; a[3] -= av; a[4] -= sv; a[5] -= sv * sv;
      }
    }
    const int mine = (in_lo ? 1 : 0) + (in_hi ? 1 : 0);
    // one reservation per wave
    int incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    const int wtot = __shfl(incl, 63, 64);
    unsigned int base = 0;
    if (lane == 0 && wtot) base = atomicAdd(counter, (unsigned int)wtot);
    base = (unsigned int)__shfl((int)base, 0, 64);
    long long pos = (long long)base + (incl - mine);
    const int64_t k = seg + (j - i - 1);
    if (in_lo && pos < shard_items) {
      fkeys[shard_base + pos] = (float)lo;
      tags[shard_base + pos] = (int)(k + 1);
    }
    pos += in_lo ? 1 : 0;
    if (in_hi && pos < shard_items) {
      fkeys[shard_base + pos] = (float)hi;
      tags[shard_base + pos] = -(int)(k + 1);
    }
  }
  hull_block_reduce<256>(a, red, rows + ((size_t)i * kHullEmitSplit + blockIdx.y) * 7);
}

// the state in front of the hull: the rows' contributions in a fixed order -> init[0..5]; init[6] = the sum of all ranges;
// more endpoints in the hull than the compacted arrays hold: the stage is repeated on the full sort
__global__ __launch_bounds__(kHullThreads) void hull_init_kernel(const double* __restrict__ rows, int nrows,
                                                                 HullPlan* __restrict__ plan, double* __restrict__ init,
                                                                 long long cap_items, const unsigned int* __restrict__ shard_count,
                                                                 int32_t* __restrict__ overflow) {
  __shared__ double tot[kHullRowThreads][7];
  __shared__ double acc7[7];
  hull_rows_total(rows, nrows, tot, acc7);
  if (threadIdx.x < 6) init[threadIdx.x] = acc7[threadIdx.x];
  if (threadIdx.x == 6) init[6] = plan->r_total;
  if (threadIdx.x == 0) {
    const long long shard_items = cap_items / kHullShards;
    long long total = 0;
    bool full = false;
    for (int k = 0; k < kHullShards; ++k) {
      const unsigned int c = shard_count[(size_t)k * kHullCountStride];
      total += c;
      full |= (long long)c > shard_items;
    }
    plan->hull_items = total;
    plan->emitted = (unsigned int)total;
    if (plan->failed || full || total == 0) {
      plan->failed = 1;
      *overflow = 1;
    }
  }
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
                                                              const ScaleSeg* __restrict__ segs,
                                                              const double* __restrict__ init = nullptr /* hull path: the
                                                              state in front of the first endpoint [0..5], sum of all ranges [6] */) {
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
    double acc = (init && q < 6) ? init[q] : 0;
    for (int k = 0; k < 1024; ++k) {
      const double v = tot[k];
      tot[k] = acc;
      acc += v;
    }
    if (q == 6) out[0] = init ? init[6] : acc;
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
    double* __restrict__ est, const ScaleSeg* __restrict__ segs, int64_t est_stride,
    int32_t* __restrict__ hull_overflow = nullptr /* hull path: raised when no endpoint of the hull has a finite cost */) {
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
    if (hull_overflow && !(c < INFINITY)) *hull_overflow = 1;
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

// hull path: the endpoint arrays hold `items` (capacity, padded with tag 0) compacted endpoints of the hull, `init` the
// state in front of it and the sum of all ranges
struct HullRun {
  int64_t items;
  const double* init;
};
hipError_t sort_and_sweep(hipStream_t s, const Work& w, const double* d_x, const double* d_r,
                          int64_t n, double* d_est, int32_t* d_overflow, const double* d_src = nullptr,
                          const double* d_dst = nullptr, int n_pts = 0, double beta = 0.0, const HullRun* hull = nullptr) {
  const int64_t m_full = 2 * n;
  const int64_t m = hull ? hull->items : m_full;
  const int64_t nblk = hull ? (m + kSwChunk - 1) / kSwChunk : w.nblk;
  const int32_t* tags = nullptr;
  const double2* sorted_xr = nullptr;
  size_t tmp = w.sort_tmp_bytes;
  if (d_overflow) {
    rocprim::double_buffer<float> kb(fkeys0(w), fkeys1(w, m_full));
    rocprim::double_buffer<int32_t> vb(w.tags[0], w.tags[1]);
    hipError_t e = rocprim::radix_sort_pairs(w.sort_tmp, tmp, kb, vb, (size_t)m, 0, 32, s);
    if (e != hipSuccess) return e;
    if (hull)
      hipLaunchKernelGGL(tls_order_fix_kernel<true>, dim3((unsigned)nblk), dim3(kSwThreads), 0, s,
                         reinterpret_cast<const uint32_t*>(kb.current()), 1, vb.current(), d_x, d_r, m, nblk, vb.alternate(),
                         w.xr_sorted, d_overflow, (int64_t)0, static_cast<const ScaleSeg*>(nullptr), d_src, d_dst, n_pts, beta);
    else
      hipLaunchKernelGGL(tls_order_fix_kernel<false>, dim3((unsigned)nblk), dim3(kSwThreads), 0, s,
                         reinterpret_cast<const uint32_t*>(kb.current()), 1, vb.current(), d_x, d_r, m, nblk, vb.alternate(),
                         w.xr_sorted, d_overflow, (int64_t)0, static_cast<const ScaleSeg*>(nullptr), d_src, d_dst, n_pts, beta);
    tags = vb.alternate();
    sorted_xr = w.xr_sorted;
    hipLaunchKernelGGL(tls_sweep_totals_kernel, dim3((unsigned)nblk), dim3(kSwThreads), 0, s, tags,
                       static_cast<const double*>(nullptr), static_cast<const double*>(nullptr), m, nblk, w.partials,
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
    hipLaunchKernelGGL(tls_sweep_totals_kernel, dim3((unsigned)nblk), dim3(kSwThreads), 0, s, tags,
                       d_x, d_r, m, nblk, w.partials, out_xr, static_cast<const ScaleSeg*>(nullptr));
  }
  hipLaunchKernelGGL(tls_sweep_scan_kernel, dim3(1, kNumAcc), dim3(1024), 0, s, w.partials, nblk,
                     w.scalars, static_cast<const ScaleSeg*>(nullptr), hull ? hull->init : static_cast<const double*>(nullptr));
  hipLaunchKernelGGL(tls_sweep_cost_kernel, dim3((unsigned)nblk), dim3(kSwThreads), 0, s, tags,
                     sorted_xr, m, nblk, w.partials, w.scalars, w.best_cost, w.best_hat,
                     w.best_pos, w.scalars + 1, static_cast<const ScaleSeg*>(nullptr));
  hipLaunchKernelGGL(tls_sweep_argmin_kernel, dim3(1), dim3(1024), 0, s, w.best_cost, w.best_hat,
                     w.best_pos, nblk, w.scalars + 1, d_est, static_cast<const ScaleSeg*>(nullptr), (int64_t)0,
                     hull ? d_overflow : static_cast<int32_t*>(nullptr));
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
    hipLaunchKernelGGL(tls_order_fix_kernel<false>, dim3((unsigned)max_nblk, (unsigned)count), dim3(kSwThreads), 0, s,
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
  const int64_t hull_pct = setting(S_SCALE_HULL);
  // (problems of up to 4096 points share one sort per batch -- solver.hip, kMidScaledN -- and a problem must give the
  // same bits alone and in a batch: the hull path is for the sizes above)
  if (d_overflow && hull_pct > 0 && n > 4096) {
    // the hull of the arg-min (above): compacted endpoints up to hull_pct % of all (a multiple of the sweep's chunk)
    const int64_t m = 2 * M;
    int64_t cap = (m / 100) * hull_pct;
    constexpr int64_t kCapUnit = (int64_t)kSwChunk * kHullShards;  // whole sweep chunks, equal shards
    cap = (cap / kCapUnit) * kCapUnit;
    // small state in keys[1] (free on the float-key path): table | histograms | prefixes | plan | rows | init
    char* q = reinterpret_cast<char*>(w.keys[1]);
    auto take = [&](size_t bytes) {
      char* r = q;
      q += align_up(bytes);
      return r;
    };
    double* T = reinterpret_cast<double*>(take(sizeof(double) * (kHullBins + 1)));
    unsigned long long* hist = reinterpret_cast<unsigned long long*>(take(8 * (size_t)kHullKinds * kHullBins));
    long long* pre = reinterpret_cast<long long*>(take(8 * 4 * (size_t)kHullBins));
    HullPlan* plan = reinterpret_cast<HullPlan*>(take(sizeof(HullPlan)));
    double* rows = reinterpret_cast<double*>(take(sizeof(double) * 7 * (size_t)n * kHullEmitSplit));
    double* init = reinterpret_cast<double*>(take(sizeof(double) * 8));
    unsigned int* shard_count = reinterpret_cast<unsigned int*>(take(sizeof(unsigned int) * kHullShards * kHullCountStride));
    if ((size_t)(q - reinterpret_cast<char*>(w.keys[1])) <= (size_t)m * 8 && cap >= kCapUnit) {
      static int cus = 0;
      if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
      }
      // histograms | table (kHullBins + 1 doubles, padded to + 2) | one 64-pair queue per wave (level 2)
      const size_t lds = 8 * (size_t)kHullKinds * kHullBins + sizeof(double) * (kHullBins + 2) + sizeof(double2) * kHullQueue * (kHullThreads / 64);
      static DynLdsOptIn optin, optin2;
      optin.ensure(reinterpret_cast<const void*>(hull_hist_kernel<false>), (int)lds);
      optin2.ensure(reinterpret_cast<const void*>(hull_hist_kernel<true>), (int)lds);
      (void)hipMemsetAsync(hist, 0, 8 * (size_t)kHullKinds * kHullBins, s);
      (void)hipMemsetAsync(shard_count, 0, sizeof(unsigned int) * kHullShards * kHullCountStride, s);
      hipLaunchKernelGGL(hull_table_kernel, dim3(1), dim3(256), 0, s, d_src, d_dst, n, T, plan);
      hipLaunchKernelGGL(hull_hist_kernel<false>, dim3((unsigned)std::min(n - 1, cus)), dim3(kHullThreads), lds, s, d_src, d_dst, n,
                         beta, T, plan, hist);
      hipLaunchKernelGGL(hull_plan_kernel, dim3(1), dim3(kHullThreads), 0, s, 0, hist, pre, T, rows, n - 1, plan, d_overflow);
      hipLaunchKernelGGL(hull_eval_kernel, dim3((unsigned)(n - 1)), dim3(256), 0, s, d_src, d_dst, n, beta, plan, rows);
      hipLaunchKernelGGL(hull_plan_kernel, dim3(1), dim3(kHullThreads), 0, s, 1, hist, pre, T, rows, n - 1, plan, d_overflow);
      bool use_hull = true;
      if (setting(S_SCALE_HULL_SYNC) != 0) {
        // ONE host look at the plan: the compacted arrays (and with them the sort, the order-fix pass and the sweep) are
        // sized by the hull itself instead of by the worst case the setting allows -- 1.6 % of the endpoints when the
        // inliers form a peak, a third on a flat top -- and a hull that failed or is too large to pay goes straight to the
        // full sort instead of through the overflow flag and a repeated solve
        HullPlan hp;
        if (hipMemcpyAsync(&hp, plan, sizeof(hp), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
          return hipErrorUnknown;
        if (!hp.failed && !hp.anomalies && hp.hull_items > m / 12 && hp.t_lo > -INFINITY && hp.t_hi < INFINITY) {
          // a wide hull (a flat top: no inlier peak): a second level of bins inside it, ten times finer, before anything
          // is sorted -- one more pass over the TRIMs for a hull a few times smaller
          (void)hipMemsetAsync(hist, 0, 8 * (size_t)kHullKinds * kHullBins, s);
          hipLaunchKernelGGL(hull_table2_kernel, dim3(1), dim3(256), 0, s, T, plan);
          hipLaunchKernelGGL(hull_hist_kernel<true>, dim3((unsigned)std::min(n - 1, cus)), dim3(kHullThreads), lds, s, d_src, d_dst, n,
                             beta, T, plan, hist);
          hipLaunchKernelGGL(hull_plan_kernel, dim3(1), dim3(kHullThreads), 0, s, 2, hist, pre, T, rows, n - 1, plan, d_overflow);
          if (hipMemcpyAsync(&hp, plan, sizeof(hp), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
            return hipErrorUnknown;
        }
        const int64_t want = hp.hull_items + hp.hull_items / 25 + kCapUnit;
        if (hp.failed || hp.anomalies || want > cap) {
          use_hull = false;
          (void)hipMemsetAsync(d_overflow, 0, sizeof(int32_t), s);
        } else {
          cap = ((want + kCapUnit - 1) / kCapUnit) * kCapUnit;
        }
      }
      if (use_hull) {
      (void)hipMemsetAsync(fkeys0(w), 0x7f, (size_t)cap * 4, s);
      (void)hipMemsetAsync(w.tags[0], 0, (size_t)cap * 4, s);
      hipLaunchKernelGGL(hull_emit_kernel, dim3((unsigned)(n - 1), kHullEmitSplit), dim3(256), 0, s, d_src, d_dst, n, beta, plan,
                         fkeys0(w), w.tags[0], rows, (long long)cap, shard_count);
      hipLaunchKernelGGL(hull_init_kernel, dim3(1), dim3(kHullThreads), 0, s, rows, (n - 1) * kHullEmitSplit, plan, init,
                         (long long)cap, shard_count, d_overflow);
      if (setting(S_K4_DEBUG)) {  // diagnostics only
        HullPlan hp;
        (void)hipStreamSynchronize(s);
        if (hipMemcpy(&hp, plan, sizeof(hp), hipMemcpyDeviceToHost) == hipSuccess)
          fprintf(stderr, "[teaser_hip] scale hull: centre %.4f, t0 %.6f, achieved cost %.6f of %.6f, level %d bins %d..%d = [%.6f, %.6f), "
                  "%lld of %lld endpoints (capacity %lld), anomalies %d, failed %d\n", hp.c, hp.t0, hp.ub, hp.r_total, hp.levels,
                  hp.first_bin, hp.last_bin, hp.t_lo, hp.t_hi, hp.hull_items, (long long)m, (long long)cap, hp.anomalies, hp.failed);
      }
      const HullRun run{cap, init};
      return sort_and_sweep(s, w, d_raw, nullptr, M, d_scale, d_overflow, d_src, d_dst, n, beta, &run);
      }
    }
  }
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
