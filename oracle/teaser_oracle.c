/*
 * teaser_oracle.c -- CPU restatement ("oracle") of the TEASER++ solve() hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see teaser_oracle.h).  Plain C11 + OpenMP, no third-party code.
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp; no -march=native, mirroring the
 * reference's default build which has no FMA: reference CMakeLists.txt:27).
 *
 * All file:line citations are relative to /root/reference.
 */
#include "teaser_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

static inline int popc64(uint64_t x) { return __builtin_popcountll(x); }
static inline int ctz64(uint64_t x) { return __builtin_ctzll(x); }

ORACLE_API void oracle_params_default(oracle_params* p) {
  /* registration.h:419-514 */
  p->noise_bound = 0.01;
  p->cbar2 = 1;
  p->estimate_scaling = 1;
  p->rotation_estimation_algorithm = 0;
  p->rotation_gnc_factor = 1.4;
  p->rotation_max_iterations = 100;
  p->rotation_cost_threshold = 1e-6;
  p->rotation_tim_graph = 0;
  p->inlier_selection_mode = 0;
  p->kcore_heuristic_threshold = 0.5;
  p->use_max_clique = 1;
  p->max_clique_exact_solution = 1;
  p->max_clique_time_limit = 3600;
#ifdef _OPENMP
  p->max_clique_num_threads = omp_get_max_threads();
#else
  p->max_clique_num_threads = 1;
#endif
}

/* ------------------------------------------------------------------------------------------- */
/* Scalar TLS -- registration.cc:21-88                                                          */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
  double v;
  int64_t tag; /* +(i+1) opening endpoint, -(i+1) closing endpoint (registration.cc:36-37) */
} tls_ep;

/* Stable merge sort by value only.  The reference uses std::sort with a value-only comparator
 * (registration.cc:41-42), which is unstable: the order of tied endpoints is unspecified there;
 * here ties keep insertion order (x-r of i, x+r of i, x-r of i+1, ...). */
static void tls_merge_sort(tls_ep* a, tls_ep* tmp, int64_t n) {
  for (int64_t width = 1; width < n; width *= 2) {
    for (int64_t lo = 0; lo < n; lo += 2 * width) {
      int64_t mid = lo + width < n ? lo + width : n;
      int64_t hi = lo + 2 * width < n ? lo + 2 * width : n;
      int64_t i = lo, j = mid, k = lo;
      while (i < mid && j < hi) tmp[k++] = (a[j].v < a[i].v) ? a[j++] : a[i++];
      while (i < mid) tmp[k++] = a[i++];
      while (j < hi) tmp[k++] = a[j++];
    }
    memcpy(a, tmp, (size_t)n * sizeof(tls_ep));
  }
}

ORACLE_API int oracle_scalar_tls(const double* X, const double* ranges, int64_t n,
                                 double* estimate, uint8_t* inliers) {
  if (n <= 0) return 1;
  int64_t nr_centers = 2 * n; /* registration.cc:47 (int there; 64-bit here) */
  tls_ep* h = (tls_ep*)malloc((size_t)nr_centers * sizeof(tls_ep));
  tls_ep* tmp = (tls_ep*)malloc((size_t)nr_centers * sizeof(tls_ep));
  if (!h || !tmp) {
    free(h);
    free(tmp);
    return 2;
  }
  for (int64_t i = 0; i < n; ++i) { /* registration.cc:35-38 */
    h[2 * i].v = X[i] - ranges[i];
    h[2 * i].tag = i + 1;
    h[2 * i + 1].v = X[i] + ranges[i];
    h[2 * i + 1].tag = -i - 1;
  }
  tls_merge_sort(h, tmp, nr_centers); /* registration.cc:41-42 */
  free(tmp);

  /* registration.cc:45-56 */
  double ranges_inverse_sum = 0;
  for (int64_t i = 0; i < n; ++i) ranges_inverse_sum += ranges[i];
  double dot_X_weights = 0, dot_weights_consensus = 0, sum_xi = 0, sum_xi_square = 0;
  int64_t consensus_set_cardinal = 0;

  double best_cost = INFINITY, best_hat = NAN;
  int have = 0;
  double first_hat = NAN;
  for (int64_t i = 0; i < nr_centers; ++i) { /* registration.cc:58-75 */
    int64_t idx = (h[i].tag > 0 ? h[i].tag : -h[i].tag) - 1;
    int epsilon = (h[i].tag > 0) ? 1 : -1;
    double w = 1.0 / (ranges[idx] * ranges[idx]); /* weights = ranges.^2 .inverse(), :45-46 */
    consensus_set_cardinal += epsilon;
    dot_weights_consensus += epsilon * w;
    dot_X_weights += epsilon * w * X[idx];
    ranges_inverse_sum -= epsilon * ranges[idx];
    sum_xi += epsilon * X[idx];
    sum_xi_square += epsilon * X[idx] * X[idx];

    double x_hat = dot_X_weights / dot_weights_consensus;
    double residual =
        consensus_set_cardinal * x_hat * x_hat + sum_xi_square - 2 * sum_xi * x_hat;
    double cost = residual + ranges_inverse_sum;
    if (i == 0) first_hat = x_hat;
    /* x_cost.minCoeff(&min_idx), registration.cc:77-78: first minimum; NaN treated as +inf */
    if (cost < best_cost) {
      best_cost = cost;
      best_hat = x_hat;
      have = 1;
    }
  }
  if (!have) best_hat = first_hat;
  free(h);
  if (estimate) *estimate = best_hat;
  if (inliers) { /* registration.cc:86 */
    for (int64_t i = 0; i < n; ++i) inliers[i] = (fabs(X[i] - best_hat) <= ranges[i]) ? 1 : 0;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* TIMs + scale stage                                                                           */
/* ------------------------------------------------------------------------------------------- */

/* computeTIMs, registration.cc:512-551: TIM k = v_j - v_i for j>i, pair order
 * k = i*N - i(i+1)/2 + (j-i-1); map(:,k) = (i,j). */
ORACLE_API int oracle_compute_tims(const double* v, int32_t n, double* tims, int32_t* map) {
  int64_t N = n;
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t i = 0; i < N - 1; ++i) {
    int64_t seg = i * N - i * (i + 1) / 2;
    for (int64_t j = i + 1; j < N; ++j) {
      int64_t k = seg + (j - i - 1);
      tims[3 * k + 0] = v[3 * j + 0] - v[3 * i + 0];
      tims[3 * k + 1] = v[3 * j + 1] - v[3 * i + 1];
      tims[3 * k + 2] = v[3 * j + 2] - v[3 * i + 2];
      if (map) {
        map[2 * k + 0] = (int32_t)i;
        map[2 * k + 1] = (int32_t)j;
      }
    }
  }
  return 0;
}

/* Column norm exactly as `array().square().colwise().sum().array().sqrt()`
 * (registration.cc:415-418, 434-437): each product and add individually rounded, sum order
 * (x^2 + y^2) + z^2, IEEE sqrt.  This translation unit is compiled with -ffp-contract=off. */
static inline double col_norm(double x, double y, double z) {
  double xx = x * x, yy = y * y, zz = z * z;
  return sqrt((xx + yy) + zz);
}

ORACLE_API int oracle_inlier_bitmap_fixed_scale(const double* src, const double* dst, int32_t n,
                                                double noise_bound, double cbar2,
                                                uint64_t* bitmap) {
  int64_t N = n, W = (N + 63) / 64;
  double beta = 2 * noise_bound * sqrt(cbar2); /* registration.cc:438 */
  memset(bitmap, 0, (size_t)(N * W) * sizeof(uint64_t));
  /* upper triangle per row (no write conflicts), then mirror */
#pragma omp parallel for schedule(dynamic, 8)
  for (int64_t i = 0; i < N; ++i) {
    double sx = src[3 * i], sy = src[3 * i + 1], sz = src[3 * i + 2];
    double dx = dst[3 * i], dy = dst[3 * i + 1], dz = dst[3 * i + 2];
    uint64_t* row = bitmap + i * W;
    for (int64_t j = i + 1; j < N; ++j) {
      double v1 = col_norm(src[3 * j] - sx, src[3 * j + 1] - sy, src[3 * j + 2] - sz);
      double v2 = col_norm(dst[3 * j] - dx, dst[3 * j + 1] - dy, dst[3 * j + 2] - dz);
      if (fabs(v1 - v2) <= beta) row[j >> 6] |= 1ull << (j & 63); /* registration.cc:442 */
    }
  }
  /* mirror: bit (j,i) = bit (i,j); rows j written by distinct threads */
#pragma omp parallel for schedule(dynamic, 8)
  for (int64_t j = 0; j < N; ++j) {
    uint64_t* row = bitmap + j * W;
    for (int64_t i = 0; i < j; ++i) {
      if ((bitmap[i * W + (j >> 6)] >> (j & 63)) & 1ull) row[i >> 6] |= 1ull << (i & 63);
    }
  }
  return 0;
}

ORACLE_API int oracle_scale_inliers_mask(const double* a, const double* b, int64_t m,
                                         double noise_bound, double cbar2,
                                         int32_t estimate_scaling, double* scale, uint8_t* mask) {
  double beta = 2 * noise_bound * sqrt(cbar2);
  if (!estimate_scaling) { /* registration.cc:427-443 */
    *scale = 1;
    for (int64_t k = 0; k < m; ++k) {
      double v1 = col_norm(a[3 * k], a[3 * k + 1], a[3 * k + 2]);
      double v2 = col_norm(b[3 * k], b[3 * k + 1], b[3 * k + 2]);
      mask[k] = fabs(v1 - v2) <= beta;
    }
    return 0;
  }
  /* registration.cc:410-425 */
  double* raw = (double*)malloc((size_t)m * sizeof(double));
  double* alphas = (double*)malloc((size_t)m * sizeof(double));
  if (!raw || !alphas) {
    free(raw);
    free(alphas);
    return 2;
  }
  for (int64_t k = 0; k < m; ++k) {
    double v1 = col_norm(a[3 * k], a[3 * k + 1], a[3 * k + 2]);
    double v2 = col_norm(b[3 * k], b[3 * k + 1], b[3 * k + 2]);
    raw[k] = v2 / v1;
    alphas[k] = beta * (1.0 / v1); /* beta * v1_dist.cwiseInverse(), :422 */
  }
  int rc = oracle_scalar_tls(raw, alphas, m, scale, mask);
  free(raw);
  free(alphas);
  return rc;
}

ORACLE_API int oracle_inlier_bitmap_tls_scale(const double* src, const double* dst, int32_t n,
                                              double noise_bound, double cbar2, double* scale,
                                              uint64_t* bitmap) {
  int64_t N = n, W = (N + 63) / 64, M = N * (N - 1) / 2;
  memset(bitmap, 0, (size_t)(N * W) * sizeof(uint64_t));
  if (M <= 0) {
    *scale = 1;
    return 0;
  }
  double beta = 2 * noise_bound * sqrt(cbar2); /* registration.cc:421 */
  double* raw = (double*)malloc((size_t)M * sizeof(double));
  double* alphas = (double*)malloc((size_t)M * sizeof(double));
  uint8_t* mask = (uint8_t*)malloc((size_t)M);
  if (!raw || !alphas || !mask) {
    free(raw);
    free(alphas);
    free(mask);
    return 2;
  }
#pragma omp parallel for schedule(dynamic, 8)
  for (int64_t i = 0; i < N - 1; ++i) {
    int64_t seg = i * N - i * (i + 1) / 2;
    for (int64_t j = i + 1; j < N; ++j) {
      int64_t k = seg + (j - i - 1);
      double v1 = col_norm(src[3 * j] - src[3 * i], src[3 * j + 1] - src[3 * i + 1],
                           src[3 * j + 2] - src[3 * i + 2]);
      double v2 = col_norm(dst[3 * j] - dst[3 * i], dst[3 * j + 1] - dst[3 * i + 1],
                           dst[3 * j + 2] - dst[3 * i + 2]);
      raw[k] = v2 / v1;
      alphas[k] = beta * (1.0 / v1);
    }
  }
  int rc = oracle_scalar_tls(raw, alphas, M, scale, mask);
  if (rc == 0) {
    for (int64_t i = 0; i < N - 1; ++i) {
      int64_t seg = i * N - i * (i + 1) / 2;
      for (int64_t j = i + 1; j < N; ++j) {
        if (mask[seg + (j - i - 1)]) {
          bitmap[i * W + (j >> 6)] |= 1ull << (j & 63);
          bitmap[j * W + (i >> 6)] |= 1ull << (i & 63);
        }
      }
    }
  }
  free(raw);
  free(alphas);
  free(mask);
  return rc;
}

/* ------------------------------------------------------------------------------------------- */
/* Maximum clique -- control flow of graph.cc:12-125, pmc replaced by an exact bitset B&B       */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
  int n, W;
  const uint64_t* adj;
  int* core;  /* core numbers (Batagelj-Zaversnik) */
  int* pos;   /* position in degeneracy (removal) order */
  int* vert;  /* degeneracy order */
  int best;   /* incumbent size (exact mode) */
  int* best_clique;
  int count_mode; /* 1: count cliques of size >= target, stop at 2 */
  int target;
  int count;
} cq_ctx;

typedef struct {
  uint64_t* P;
  int* order;
  int* colour;
  int cap;
} cq_level;

typedef struct {
  cq_ctx* ctx;
  cq_level* lv;
  int nlv;
  uint64_t *Q, *Qc;
  int* C;
} cq_thread;

static void cq_level_reserve(cq_thread* T, int depth, int cap) {
  if (depth >= T->nlv) {
    int nn = T->nlv ? T->nlv * 2 : 16;
    while (nn <= depth) nn *= 2;
    T->lv = (cq_level*)realloc(T->lv, (size_t)nn * sizeof(cq_level));
    memset(T->lv + T->nlv, 0, (size_t)(nn - T->nlv) * sizeof(cq_level));
    T->nlv = nn;
  }
  cq_level* L = &T->lv[depth];
  if (!L->P) L->P = (uint64_t*)malloc((size_t)T->ctx->W * sizeof(uint64_t));
  if (L->cap < cap) {
    free(L->order);
    free(L->colour);
    L->order = (int*)malloc((size_t)cap * sizeof(int));
    L->colour = (int*)malloc((size_t)cap * sizeof(int));
    L->cap = cap;
  }
}

static void cq_thread_free(cq_thread* T) {
  for (int d = 0; d < T->nlv; ++d) {
    free(T->lv[d].P);
    free(T->lv[d].order);
    free(T->lv[d].colour);
  }
  free(T->lv);
  free(T->Q);
  free(T->Qc);
  free(T->C);
}

static inline int cq_need(const cq_ctx* ctx, int csize) {
  /* we look for a clique in P of size strictly greater than `need` */
  int b;
  if (ctx->count_mode) return ctx->target - csize - 1;
#pragma omp atomic read
  b = ctx->best;
  return b - csize;
}

/* Greedy sequential bitset colouring (BBMC style).  Lists only vertices whose colour exceeds
 * `need`, in non-decreasing colour order; returns the number listed.  Early exit (return 0)
 * as soon as colours_used + uncoloured <= need. */
static int cq_colour_sort(cq_thread* T, const uint64_t* P, int pcount, int need, int* order,
                          int* colour) {
  const cq_ctx* ctx = T->ctx;
  const int W = ctx->W;
  uint64_t *Q = T->Q, *Qc = T->Qc;
  memcpy(Q, P, (size_t)W * sizeof(uint64_t));
  int remaining = pcount, k = 0, cnt = 0;
  while (remaining > 0) {
    if (k + remaining <= need) return cnt; /* cnt is 0 here: nothing can exceed need */
    ++k;
    memcpy(Qc, Q, (size_t)W * sizeof(uint64_t));
    for (int w = 0; w < W; ++w) {
      while (Qc[w]) {
        int b = ctz64(Qc[w]);
        int v = w * 64 + b;
        const uint64_t* av = ctx->adj + (size_t)v * W;
        Qc[w] &= ~(1ull << b);
        Q[w] &= ~(1ull << b);
        for (int x = w; x < W; ++x) Qc[x] &= ~av[x];
        --remaining;
        if (k > need) {
          order[cnt] = v;
          colour[cnt] = k;
          ++cnt;
        }
      }
    }
  }
  return cnt;
}

static void cq_record(cq_thread* T, int csize) {
  cq_ctx* ctx = T->ctx;
#pragma omp critical(cq_best)
  {
    if (ctx->count_mode) {
      if (csize >= ctx->target) ctx->count++;
    } else if (csize > ctx->best) {
      memcpy(ctx->best_clique, T->C, (size_t)csize * sizeof(int));
#pragma omp atomic write
      ctx->best = csize;
    }
  }
}

static int cq_stop(const cq_ctx* ctx) {
  int c;
  if (!ctx->count_mode) return 0;
#pragma omp atomic read
  c = ctx->count;
  return c >= 2;
}

static void cq_expand(cq_thread* T, int depth, int csize, int pcount) {
  cq_ctx* ctx = T->ctx;
  const int W = ctx->W;
  cq_level_reserve(T, depth, pcount);
  cq_level_reserve(T, depth + 1, 0);
  cq_level* L = &T->lv[depth];
  int m = cq_colour_sort(T, L->P, pcount, cq_need(ctx, csize), L->order, L->colour);
  for (int i = m - 1; i >= 0; --i) {
    if (cq_stop(ctx)) return;
    if (L->colour[i] <= cq_need(ctx, csize)) return;
    int v = L->order[i];
    T->C[csize] = v;
    cq_level_reserve(T, depth + 1, 0);
    L = &T->lv[depth]; /* lv may have been realloc'ed */
    uint64_t* NP = T->lv[depth + 1].P;
    const uint64_t* av = ctx->adj + (size_t)v * W;
    int cnt = 0;
    for (int w = 0; w < W; ++w) {
      NP[w] = L->P[w] & av[w];
      cnt += popc64(NP[w]);
    }
    if (cnt == 0) {
      cq_record(T, csize + 1);
    } else if (cnt > cq_need(ctx, csize + 1)) {
      cq_expand(T, depth + 1, csize + 1, cnt);
      L = &T->lv[depth];
    }
    L->P[v >> 6] &= ~(1ull << (v & 63));
  }
}

/* Batagelj-Zaversnik O(E) core decomposition (what pmc::pmc_graph::compute_cores provides at
 * graph.cc:58-59).  Fills core[], vert[] (degeneracy order), pos[]. Returns max core. */
static int cq_cores(const uint64_t* adj, int n, int W, int* core, int* vert, int* pos) {
  int* deg = (int*)malloc((size_t)n * sizeof(int));
  int md = 0;
  for (int v = 0; v < n; ++v) {
    int d = 0;
    for (int w = 0; w < W; ++w) d += popc64(adj[(size_t)v * W + w]);
    deg[v] = d;
    if (d > md) md = d;
  }
  int* bin = (int*)calloc((size_t)md + 2, sizeof(int));
  for (int v = 0; v < n; ++v) bin[deg[v]]++;
  int start = 0;
  for (int d = 0; d <= md; ++d) {
    int num = bin[d];
    bin[d] = start;
    start += num;
  }
  for (int v = 0; v < n; ++v) {
    pos[v] = bin[deg[v]];
    vert[pos[v]] = v;
    bin[deg[v]]++;
  }
  for (int d = md; d >= 1; --d) bin[d] = bin[d - 1];
  bin[0] = 0;
  int max_core = 0;
  for (int i = 0; i < n; ++i) {
    int v = vert[i];
    core[v] = deg[v];
    if (core[v] > max_core) max_core = core[v];
    const uint64_t* av = adj + (size_t)v * W;
    for (int w = 0; w < W; ++w) {
      uint64_t bits = av[w];
      while (bits) {
        int u = w * 64 + ctz64(bits);
        bits &= bits - 1;
        if (deg[u] > deg[v]) {
          int du = deg[u], pu = pos[u], pw = bin[du], x = vert[pw];
          if (u != x) {
            pos[u] = pw;
            vert[pu] = x;
            pos[x] = pu;
            vert[pw] = u;
          }
          bin[du]++;
          deg[u]--;
        }
      }
    }
  }
  free(deg);
  free(bin);
  return max_core;
}

/* Greedy heuristic clique (stands in for pmc::pmc_heu::search, graph.cc:88-91): from `start`,
 * repeatedly add every candidate adjacent to all other candidates, else the candidate with the
 * most neighbours among the candidates (ties: smallest index). */
static int cq_greedy(const uint64_t* adj, int n, int W, int start, const int* core, int min_core,
                     uint64_t* P, int* clique) {
  int size = 0;
  clique[size++] = start;
  const uint64_t* as = adj + (size_t)start * W;
  int pc = 0;
  for (int w = 0; w < W; ++w) {
    uint64_t bits = as[w], keep = 0;
    while (bits) {
      int b = ctz64(bits);
      bits &= bits - 1;
      if (core[w * 64 + b] >= min_core) keep |= 1ull << b;
    }
    P[w] = keep;
    pc += popc64(keep);
  }
  while (pc > 0) {
    int best_v = -1, best_d = -1, added = 0;
    for (int w = 0; w < W; ++w) {
      uint64_t bits = P[w];
      while (bits) {
        int u = w * 64 + ctz64(bits);
        bits &= bits - 1;
        const uint64_t* au = adj + (size_t)u * W;
        int d = 0;
        for (int x = 0; x < W; ++x) d += popc64(au[x] & P[x]);
        if (d == pc - 1) { /* adjacent to every other candidate: in every maximal extension */
          clique[size++] = u;
          ++added;
        } else if (d > best_d) {
          best_d = d;
          best_v = u;
        }
      }
    }
    if (added) {
      /* remove the universal vertices from P (P stays a subset of each one's neighbourhood) */
      for (int i = size - added; i < size; ++i) P[clique[i] >> 6] &= ~(1ull << (clique[i] & 63));
      pc -= added;
      continue;
    }
    clique[size++] = best_v;
    const uint64_t* av = adj + (size_t)best_v * W;
    pc = 0;
    for (int w = 0; w < W; ++w) {
      P[w] &= av[w];
      pc += popc64(P[w]);
    }
  }
  return size;
}

static int cmp_int(const void* a, const void* b) {
  int x = *(const int*)a, y = *(const int*)b;
  return (x > y) - (x < y);
}

/* Run the B&B over all roots.  Exact mode: improves ctx->best/best_clique.  Count mode: counts
 * maximum cliques (each rooted at its earliest vertex in degeneracy order), stops at 2. */
static void cq_search(cq_ctx* ctx, int num_threads) {
  const int n = ctx->n, W = ctx->W;
#ifdef _OPENMP
  if (num_threads <= 0) num_threads = omp_get_max_threads();
#else
  num_threads = 1;
#endif
#pragma omp parallel num_threads(num_threads)
  {
    cq_thread T;
    memset(&T, 0, sizeof(T));
    T.ctx = ctx;
    T.Q = (uint64_t*)malloc((size_t)W * sizeof(uint64_t));
    T.Qc = (uint64_t*)malloc((size_t)W * sizeof(uint64_t));
    T.C = (int*)malloc((size_t)(n + 1) * sizeof(int));
#pragma omp for schedule(dynamic, 1)
    for (int i = 0; i < n; ++i) {
      if (cq_stop(ctx)) continue;
      int v = ctx->vert[i];
      int need = cq_need(ctx, 1); /* want clique in P bigger than need */
      int min_core = ctx->count_mode ? ctx->target - 1 : need + 1;
      if (ctx->core[v] < min_core) continue;
      cq_level_reserve(&T, 0, 0);
      uint64_t* P = T.lv[0].P;
      const uint64_t* av = ctx->adj + (size_t)v * W;
      int pc = 0;
      for (int w = 0; w < W; ++w) {
        uint64_t bits = av[w], keep = 0;
        while (bits) {
          int b = ctz64(bits);
          bits &= bits - 1;
          int u = w * 64 + b;
          if (ctx->pos[u] > i && ctx->core[u] >= min_core) keep |= 1ull << b;
        }
        P[w] = keep;
        pc += popc64(keep);
      }
      T.C[0] = v;
      if (pc == 0) {
        if (ctx->count_mode ? (1 >= ctx->target) : 0) cq_record(&T, 1);
        continue;
      }
      if (pc <= need) continue;
      cq_expand(&T, 0, 1, pc);
    }
    cq_thread_free(&T);
  }
}

ORACLE_API int oracle_max_clique(const uint64_t* bitmap, int32_t n, int32_t mode,
                                 int32_t num_threads, int32_t* clique_out, int32_t* clique_size,
                                 int32_t* unique, int32_t* max_core_out, int32_t* exact_run) {
  int W = (n + 63) / 64;
  if (unique) *unique = 0;
  if (exact_run) *exact_run = 0;
  if (max_core_out) *max_core_out = 0;
  *clique_size = 0;
  if (n <= 0) return 0;
  cq_ctx ctx;
  memset(&ctx, 0, sizeof(ctx));
  ctx.n = n;
  ctx.W = W;
  ctx.adj = bitmap;
  ctx.core = (int*)malloc((size_t)n * sizeof(int));
  ctx.pos = (int*)malloc((size_t)n * sizeof(int));
  ctx.vert = (int*)malloc((size_t)n * sizeof(int));
  ctx.best_clique = (int*)malloc((size_t)(n + 1) * sizeof(int));
  /* graph.cc:58-59, 83-85: ub = max_core + 1 */
  int max_core = cq_cores(bitmap, n, W, ctx.core, ctx.vert, ctx.pos);
  if (max_core_out) *max_core_out = max_core;
  int ub = max_core + 1;

  /* graph.cc:88-91: heuristic lower bound.  Greedy from the (up to) 32 latest vertices of the
   * degeneracy order (= highest cores). */
  {
    uint64_t* P = (uint64_t*)malloc((size_t)W * sizeof(uint64_t));
    int* tmp = (int*)malloc((size_t)(n + 1) * sizeof(int));
    ctx.best = 1;
    ctx.best_clique[0] = ctx.vert[n - 1];
    int tries = n < 32 ? n : 32;
    for (int s = 0; s < tries && ctx.best < ub; ++s) {
      int v = ctx.vert[n - 1 - s];
      if (ctx.core[v] < ctx.best) break;
      int sz = cq_greedy(bitmap, n, W, v, ctx.core, ctx.best, P, tmp);
      if (sz > ctx.best) {
        ctx.best = sz;
        memcpy(ctx.best_clique, tmp, (size_t)sz * sizeof(int));
      }
    }
    free(P);
    free(tmp);
  }

  /* graph.cc:100-122: done if lb == ub, or heuristic-only mode; else exact search */
  if (ctx.best != ub && mode == 0) {
    if (exact_run) *exact_run = 1;
    ctx.count_mode = 0;
    cq_search(&ctx, num_threads);
  }
  int size = ctx.best;
  memcpy(clique_out, ctx.best_clique, (size_t)size * sizeof(int));
  qsort(clique_out, (size_t)size, sizeof(int), cmp_int); /* registration.cc:636 */
  *clique_size = size;

  if (unique && mode == 0) {
    ctx.count_mode = 1;
    ctx.target = size;
    ctx.count = 0;
    cq_search(&ctx, num_threads);
    *unique = (ctx.count == 1);
  }
  free(ctx.core);
  free(ctx.pos);
  free(ctx.vert);
  free(ctx.best_clique);
  return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* svdRot -- utils.h:121-136                                                                    */
/* ------------------------------------------------------------------------------------------- */
static double det3(const double* M) { /* row-major */
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
         M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* One-sided (Hestenes) Jacobi SVD of a 3x3: H = U diag(S) V^T, S descending.  Stands in for
 * Eigen::JacobiSVD<Matrix3d> (utils.h:127); any correct SVD gives the same V diag(1,1,d) U^T
 * when sigma_2 + d sigma_3 > 0. All matrices row-major. */
static void svd3(const double* H, double* U, double* S, double* V) {
  double B[9];
  memcpy(B, H, sizeof(B));
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    int rotated = 0;
    for (int p = 0; p < 2; ++p) {
      for (int q = p + 1; q < 3; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < 3; ++r) {
          alpha += B[3 * r + p] * B[3 * r + p];
          beta += B[3 * r + q] * B[3 * r + q];
          gamma += B[3 * r + p] * B[3 * r + q];
        }
        if (gamma == 0.0 || fabs(gamma) <= 1e-17 * sqrt(alpha * beta)) continue;
        rotated = 1;
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int r = 0; r < 3; ++r) {
          double bp = B[3 * r + p], bq = B[3 * r + q];
          B[3 * r + p] = c * bp - s * bq;
          B[3 * r + q] = s * bp + c * bq;
          double vp = V[3 * r + p], vq = V[3 * r + q];
          V[3 * r + p] = c * vp - s * vq;
          V[3 * r + q] = s * vp + c * vq;
        }
      }
    }
    if (!rotated) break;
  }
  double sig[3];
  int idx[3] = {0, 1, 2};
  for (int c = 0; c < 3; ++c)
    sig[c] = sqrt(B[c] * B[c] + B[3 + c] * B[3 + c] + B[6 + c] * B[6 + c]);
  for (int a = 0; a < 2; ++a)
    for (int b = a + 1; b < 3; ++b)
      if (sig[idx[b]] > sig[idx[a]]) {
        int t = idx[a];
        idx[a] = idx[b];
        idx[b] = t;
      }
  double Bs[9], Vs[9];
  for (int c = 0; c < 3; ++c) {
    S[c] = sig[idx[c]];
    for (int r = 0; r < 3; ++r) {
      Bs[3 * r + c] = B[3 * r + idx[c]];
      Vs[3 * r + c] = V[3 * r + idx[c]];
    }
  }
  memcpy(V, Vs, sizeof(Vs));
  /* U columns */
  double tiny = 1e-300;
  double u0[3], u1[3], u2[3];
  if (S[0] > tiny) {
    for (int r = 0; r < 3; ++r) u0[r] = Bs[3 * r] / S[0];
  } else {
    u0[0] = 1;
    u0[1] = 0;
    u0[2] = 0;
  }
  if (S[1] > tiny && S[1] > 1e-15 * S[0]) {
    for (int r = 0; r < 3; ++r) u1[r] = Bs[3 * r + 1] / S[1];
  } else {
    /* any unit vector orthogonal to u0 */
    int k = (fabs(u0[0]) <= fabs(u0[1]) && fabs(u0[0]) <= fabs(u0[2])) ? 0
            : (fabs(u0[1]) <= fabs(u0[2]) ? 1 : 2);
    double e[3] = {0, 0, 0};
    e[k] = 1;
    double d = u0[k];
    double nn = 0;
    for (int r = 0; r < 3; ++r) {
      u1[r] = e[r] - d * u0[r];
      nn += u1[r] * u1[r];
    }
    nn = sqrt(nn);
    for (int r = 0; r < 3; ++r) u1[r] /= nn;
  }
  if (S[2] > tiny && S[2] > 1e-15 * S[0]) {
    for (int r = 0; r < 3; ++r) u2[r] = Bs[3 * r + 2] / S[2];
  } else {
    u2[0] = u0[1] * u1[2] - u0[2] * u1[1];
    u2[1] = u0[2] * u1[0] - u0[0] * u1[2];
    u2[2] = u0[0] * u1[1] - u0[1] * u1[0];
  }
  for (int r = 0; r < 3; ++r) {
    U[3 * r] = u0[r];
    U[3 * r + 1] = u1[r];
    U[3 * r + 2] = u2[r];
  }
}

static void svd_rot_from_H(const double* H, double* R) {
  double U[9], S[3], V[9];
  svd3(H, U, S, V);
  if (det3(U) * det3(V) < 0) { /* utils.h:131-133 */
    V[2] = -V[2];
    V[5] = -V[5];
    V[8] = -V[8];
  }
  for (int r = 0; r < 3; ++r) /* R = V U^T, utils.h:135 */
    for (int c = 0; c < 3; ++c)
      R[3 * r + c] = V[3 * r] * U[3 * c] + V[3 * r + 1] * U[3 * c + 1] + V[3 * r + 2] * U[3 * c + 2];
}

ORACLE_API void oracle_svd_rot(const double* X, const double* Y, const double* Wt, int64_t k,
                               double* R) {
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; /* H = X diag(W) Y^T, utils.h:125 */
  for (int64_t j = 0; j < k; ++j) {
    double w = Wt ? Wt[j] : 1.0;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) H[3 * r + c] += X[3 * j + r] * w * Y[3 * j + c];
  }
  svd_rot_from_H(H, R);
}

/* ------------------------------------------------------------------------------------------- */
/* GNC-TLS rotation -- registration.cc:764-866                                                  */
/* ------------------------------------------------------------------------------------------- */
ORACLE_API int oracle_gnc_tls_rotation(const double* src, const double* dst, int64_t k,
                                       double noise_bound, double gnc_factor,
                                       int64_t max_iterations, double cost_threshold, double* R,
                                       uint8_t* inliers, double* cost_out, int32_t* iterations) {
  double mu = 1;
  double prev_cost = INFINITY, cost = INFINITY;
  double noise_bound_sq = noise_bound * noise_bound; /* std::pow(nb, 2), :793 */
  if (noise_bound_sq < 1e-16) noise_bound_sq = 1e-2; /* :794-796 */
  double* weights = (double*)malloc((size_t)(k > 0 ? k : 1) * sizeof(double));
  double* res = (double*)malloc((size_t)(k > 0 ? k : 1) * sizeof(double));
  for (int64_t j = 0; j < k; ++j) weights[j] = 1.0;
  int32_t iters = 0;
  for (int64_t i = 0; i < max_iterations; ++i) {
    ++iters;
    oracle_svd_rot(src, dst, weights, k, R); /* :809 */
    double max_res = -INFINITY;
    for (int64_t j = 0; j < k; ++j) { /* :812-813 */
      double s = 0;
      for (int r = 0; r < 3; ++r) {
        double d = dst[3 * j + r] - (R[3 * r] * src[3 * j] + R[3 * r + 1] * src[3 * j + 1] +
                                     R[3 * r + 2] * src[3 * j + 2]);
        s += d * d;
      }
      res[j] = s;
      if (s > max_res) max_res = s;
    }
    if (i == 0) { /* :814-825 */
      mu = 1 / (2 * max_res / noise_bound_sq - 1);
      if (mu <= 0) break;
    }
    double th1 = (mu + 1) / mu * noise_bound_sq; /* :828-829 */
    double th2 = mu / (mu + 1) * noise_bound_sq;
    cost = 0;
    for (int64_t j = 0; j < k; ++j) { /* :831-844 */
      cost += weights[j] * res[j];
      if (res[j] >= th1) {
        weights[j] = 0;
      } else if (res[j] <= th2) {
        weights[j] = 1;
      } else {
        weights[j] = sqrt(noise_bound_sq * mu * (mu + 1) / res[j]) - mu;
      }
    }
    double cost_diff = fabs(cost - prev_cost); /* :847-858 */
    mu = mu * gnc_factor;
    prev_cost = cost;
    if (cost_diff < cost_threshold) break;
  }
  if (inliers)
    for (int64_t j = 0; j < k; ++j) inliers[j] = weights[j] >= 0.5; /* :861-865 */
  if (cost_out) *cost_out = cost;
  if (iterations) *iterations = iters;
  free(weights);
  free(res);
  return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* FGR rotation -- registration.cc:206-278                                                      */
/* ------------------------------------------------------------------------------------------- */
/* utils::calculateDiameter, utils.h:107-112.  NB: the reference function RETURNS float, so the
 * diameter is rounded to single precision before it enters the (double) mu schedule. */
static double calc_diameter3(const double* X, int64_t k) {
  double cog[3] = {0, 0, 0};
  for (int64_t j = 0; j < k; ++j)
    for (int r = 0; r < 3; ++r) cog[r] += X[3 * j + r];
  for (int r = 0; r < 3; ++r) cog[r] = cog[r] / (double)k;
  double mx = -INFINITY;
  for (int64_t j = 0; j < k; ++j) {
    double s = 0;
    for (int r = 0; r < 3; ++r) {
      double d = X[3 * j + r] - cog[r];
      s += d * d;
    }
    if (s > mx) mx = s;
  }
  return (double)(float)(2 * sqrt(mx));
}

ORACLE_API int oracle_fgr_rotation(const double* src, const double* dst, int64_t k,
                                   double noise_bound, double gnc_factor, int64_t max_iterations,
                                   double cost_threshold, double* R, uint8_t* inliers,
                                   double* cost_out, int32_t* iterations) {
  double noise_bound_sq = noise_bound * noise_bound; /* :221 */
  double cost = INFINITY;                            /* :223 */
  double src_diameter = calc_diameter3(src, k);      /* :226-227 */
  double dest_diameter = calc_diameter3(dst, k);
  double global_scale = src_diameter > dest_diameter ? src_diameter : dest_diameter; /* :228 */
  global_scale /= noise_bound_sq;                                                    /* :229 */
  double mu = global_scale * global_scale / noise_bound_sq;                          /* :230 */
  const double min_mu = 1.0;                                                         /* :233 */
  for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;                       /* :234 */
  double* l_pq = (double*)malloc((size_t)(k > 0 ? k : 1) * sizeof(double));
  for (int64_t j = 0; j < k; ++j) l_pq[j] = 1.0; /* :235-236 */
  int32_t iters = 0;
  for (int64_t i = 0; i < max_iterations; ++i) { /* :242 */
    ++iters;
    double scaled_mu = mu * noise_bound_sq; /* :243 */
    for (int64_t j = 0; j < k; ++j) {       /* :247-253 */
      double s = 0;
      for (int r = 0; r < 3; ++r) {
        double d = dst[3 * j + r] - (R[3 * r] * src[3 * j] + R[3 * r + 1] * src[3 * j + 1] +
                                     R[3 * r + 2] * src[3 * j + 2]);
        s += d * d;
      }
      double q = scaled_mu / (scaled_mu + s);
      l_pq[j] = q * q;
    }
    oracle_svd_rot(src, dst, l_pq, k, R); /* :256 */
    cost = 0;                             /* :259-262 */
    for (int64_t j = 0; j < k; ++j) {
      double s = 0;
      for (int r = 0; r < 3; ++r) {
        double d = dst[3 * j + r] - (R[3 * r] * src[3 * j] + R[3 * r + 1] * src[3 * j + 1] +
                                     R[3 * r + 2] * src[3 * j + 2]);
        s += d * d;
      }
      cost += (scaled_mu * s) / (scaled_mu + s);
    }
    if (cost < cost_threshold || mu < min_mu) break; /* :265-271 */
    mu /= gnc_factor;                                /* :274 */
  }
  if (inliers)
    for (int64_t j = 0; j < k; ++j) inliers[j] = l_pq[j] != 0.0; /* l_pq.cast<bool>(), :277-279 */
  if (cost_out) *cost_out = cost;
  if (iterations) *iterations = iters;
  free(l_pq);
  return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* Quatro rotation (yaw only) -- registration.cc:280-408, utils::svdRot2d utils.h:145-160       */
/* ------------------------------------------------------------------------------------------- */
/* 2x2 SVD rotation.  R = V diag(1, det(U)det(V)) U^T maximises tr(R H) over SO(2); for a 2x2 H
 * that maximiser is the rotation by atan2(H01 - H10, H00 + H11) (unique unless both vanish, where
 * the reference's JacobiSVD output is implementation-defined; identity is returned here). */
static void svd_rot2d(const double* X, const double* Y, const double* Wt, int64_t k, double* R2) {
  double h00 = 0, h01 = 0, h10 = 0, h11 = 0; /* H = X diag(W) Y^T over the top two rows */
  for (int64_t j = 0; j < k; ++j) {
    double w = Wt[j];
    h00 += X[3 * j] * w * Y[3 * j];
    h01 += X[3 * j] * w * Y[3 * j + 1];
    h10 += X[3 * j + 1] * w * Y[3 * j];
    h11 += X[3 * j + 1] * w * Y[3 * j + 1];
  }
  double a = h00 + h11, b = h01 - h10, nrm = sqrt(a * a + b * b);
  double c = 1, sn = 0;
  if (nrm > 0) {
    c = a / nrm;
    sn = b / nrm;
  }
  R2[0] = c;
  R2[1] = -sn;
  R2[2] = sn;
  R2[3] = c;
}

ORACLE_API int oracle_quatro_rotation(const double* src, const double* dst, int64_t k,
                                      double noise_bound, double gnc_factor,
                                      int64_t max_iterations, double cost_threshold, double* R,
                                      uint8_t* inliers, double* cost_out, int32_t* iterations) {
  /* NB registration.cc:329-330 keeps the noise bound in function-local statics (first call in the
   * process wins); restated with the per-call value, which is what a fresh process sees. */
  double mu = 1; /* :324 */
  double prev_cost = INFINITY, cost = INFINITY;
  double noise_bound_sq = noise_bound * noise_bound;
  if (noise_bound_sq < 1e-16) noise_bound_sq = 1e-2; /* :331-333 */
  double* weights = (double*)malloc((size_t)(k > 0 ? k : 1) * sizeof(double));
  double* res = (double*)malloc((size_t)(k > 0 ? k : 1) * sizeof(double));
  for (int64_t j = 0; j < k; ++j) weights[j] = 1.0;
  double R2[4] = {1, 0, 0, 1};
  int32_t iters = 0;
  for (int64_t i = 0; i < max_iterations; ++i) { /* :343 */
    ++iters;
    svd_rot2d(src, dst, weights, k, R2); /* :346 */
    double max_res = -INFINITY;
    for (int64_t j = 0; j < k; ++j) { /* :349-350 */
      double d0 = dst[3 * j] - (R2[0] * src[3 * j] + R2[1] * src[3 * j + 1]);
      double d1 = dst[3 * j + 1] - (R2[2] * src[3 * j] + R2[3] * src[3 * j + 1]);
      res[j] = d0 * d0 + d1 * d1;
      if (res[j] > max_res) max_res = res[j];
    }
    if (i == 0) { /* :351-362 */
      mu = 1 / (2 * max_res / noise_bound_sq - 1);
      if (mu <= 0) break;
    }
    double th1 = (mu + 1) / mu * noise_bound_sq; /* :365-366 */
    double th2 = mu / (mu + 1) * noise_bound_sq;
    cost = 0;
    for (int64_t j = 0; j < k; ++j) { /* :368-381 */
      cost += weights[j] * res[j];
      if (res[j] >= th1) {
        weights[j] = 0;
      } else if (res[j] <= th2) {
        weights[j] = 1;
      } else {
        weights[j] = sqrt(noise_bound_sq * mu * (mu + 1) / res[j]) - mu;
      }
    }
    double cost_diff = fabs(cost - prev_cost); /* :384-395 */
    mu = mu * gnc_factor;
    prev_cost = cost;
    if (cost_diff < cost_threshold) break;
  }
  if (inliers)
    for (int64_t j = 0; j < k; ++j) inliers[j] = weights[j] >= 0.4; /* :398-402 */
  for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;       /* :292, :407 */
  R[0] = R2[0];
  R[1] = R2[1];
  R[3] = R2[2];
  R[4] = R2[3];
  if (cost_out) *cost_out = cost;
  if (iterations) *iterations = iters;
  free(weights);
  free(res);
  return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* TLS translation -- registration.cc:445-471                                                   */
/* ------------------------------------------------------------------------------------------- */
ORACLE_API int oracle_tls_translation(const double* src, const double* dst, int64_t k,
                                      double noise_bound, double cbar2, double* t,
                                      uint8_t* inliers) {
  if (k <= 0) return 1;
  double beta = noise_bound * sqrt(cbar2); /* :459 */
  double* raw = (double*)malloc((size_t)k * sizeof(double));
  double* alphas = (double*)malloc((size_t)k * sizeof(double));
  uint8_t* tmp = (uint8_t*)malloc((size_t)k);
  for (int64_t j = 0; j < k; ++j) {
    alphas[j] = beta * 1.0;
    if (inliers) inliers[j] = 1;
  }
  for (int a = 0; a < 3; ++a) { /* :465-470 */
    for (int64_t j = 0; j < k; ++j) raw[j] = dst[3 * j + a] - src[3 * j + a]; /* :455 */
    oracle_scalar_tls(raw, alphas, k, &t[a], tmp);
    if (inliers)
      for (int64_t j = 0; j < k; ++j) inliers[j] = inliers[j] && tmp[j];
  }
  free(raw);
  free(alphas);
  free(tmp);
  return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* solve -- registration.cc:568-737                                                             */
/* ------------------------------------------------------------------------------------------- */
ORACLE_API int oracle_solve(const oracle_params* p, const double* src, const double* dst,
                            int32_t n, oracle_solution* sol, int32_t* clique_out,
                            int32_t* rot_inliers_out, int64_t rot_capacity,
                            int32_t* trans_inliers_out) {
  memset(sol, 0, sizeof(*sol));
  sol->valid = 1; /* registration.h:33 */
  for (int i = 0; i < 3; ++i) sol->rotation[4 * i] = 1.0;
  if (p->rotation_estimation_algorithm < 0 || p->rotation_estimation_algorithm > 2) return 3;
  int64_t N = n, W = (N + 63) / 64;
  int mode = p->inlier_selection_mode;
  if (!p->use_max_clique) mode = 3;            /* registration.cc:574-578 */
  if (!p->max_clique_exact_solution) mode = 1; /* registration.cc:579-583 */

  uint64_t* bitmap = (uint64_t*)malloc((size_t)(N * W > 0 ? N * W : 1) * sizeof(uint64_t));
  int* clique = (int*)malloc((size_t)(N + 1) * sizeof(int));
  if (!bitmap || !clique) return 2;
  int K = 0;

  /* registration.cc:599-603 */
  if (p->estimate_scaling) {
    oracle_inlier_bitmap_tls_scale(src, dst, n, p->noise_bound, p->cbar2, &sol->scale, bitmap);
  } else {
    sol->scale = 1;
    oracle_inlier_bitmap_fixed_scale(src, dst, n, p->noise_bound, p->cbar2, bitmap);
  }
  {
    int64_t e = 0;
    for (int64_t i = 0; i < N * W; ++i) e += popc64(bitmap[i]);
    sol->num_edges = e / 2;
  }

  /* registration.cc:609-654 */
  if (mode != 3) {
    int cmode = (mode == 0) ? 0 : 1; /* KCORE_HEU is unpinned: heuristic clique */
    oracle_max_clique(bitmap, n, cmode, p->max_clique_num_threads, clique, &K,
                      &sol->clique_unique, &sol->max_core, &sol->clique_exact_run);
    sol->clique_size = K;
    if (clique_out) memcpy(clique_out, clique, (size_t)K * sizeof(int));
    if (K <= 1) { /* :643-647 */
      sol->valid = 0;
      free(bitmap);
      free(clique);
      return 0;
    }
  } else {
    K = n;
    for (int i = 0; i < n; ++i) clique[i] = i;
    sol->clique_size = K;
    sol->clique_unique = 1;
    if (clique_out) memcpy(clique_out, clique, (size_t)K * sizeof(int));
  }
  free(bitmap);

  /* registration.cc:657-694: TIMs for rotation */
  int64_t KT;
  double *ps, *pd;
  if (p->rotation_tim_graph == 0) { /* CHAIN */
    KT = K;
    ps = (double*)malloc((size_t)(3 * KT) * sizeof(double));
    pd = (double*)malloc((size_t)(3 * KT) * sizeof(double));
    for (int i = 0; i < K; ++i) {
      int root = clique[i];
      int leaf = (i != K - 1) ? clique[i + 1] : clique[0];
      for (int r = 0; r < 3; ++r) {
        ps[3 * i + r] = src[3 * (int64_t)leaf + r] - src[3 * (int64_t)root + r];
        pd[3 * i + r] = dst[3 * (int64_t)leaf + r] - dst[3 * (int64_t)root + r];
      }
    }
  } else { /* COMPLETE */
    KT = (int64_t)K * (K - 1) / 2;
    ps = (double*)malloc((size_t)(3 * KT + 3) * sizeof(double));
    pd = (double*)malloc((size_t)(3 * KT + 3) * sizeof(double));
    double* si = (double*)malloc((size_t)(3 * K) * sizeof(double));
    double* di = (double*)malloc((size_t)(3 * K) * sizeof(double));
    for (int i = 0; i < K; ++i)
      for (int r = 0; r < 3; ++r) {
        si[3 * i + r] = src[3 * (int64_t)clique[i] + r];
        di[3 * i + r] = dst[3 * (int64_t)clique[i] + r];
      }
    oracle_compute_tims(si, K, ps, NULL);
    oracle_compute_tims(di, K, pd, NULL);
    free(si);
    free(di);
  }
  /* :697 pruned_dst_tims_ *= (1 / scale) */
  double inv_scale = 1 / sol->scale;
  for (int64_t i = 0; i < 3 * KT; ++i) pd[i] *= inv_scale;
  /* :702-704 rotation noise bound *= 2 / scale */
  double rot_nb = p->noise_bound * (2 / sol->scale);

  uint8_t* rmask = (uint8_t*)malloc((size_t)(KT > 0 ? KT : 1));
  /* registration.h:856-869: the rotation estimator chosen by rotation_estimation_algorithm */
  if (p->rotation_estimation_algorithm == 1)
    oracle_fgr_rotation(ps, pd, KT, rot_nb, p->rotation_gnc_factor, p->rotation_max_iterations,
                        p->rotation_cost_threshold, sol->rotation, rmask, &sol->gnc_cost,
                        &sol->gnc_iterations);
  else if (p->rotation_estimation_algorithm == 2)
    oracle_quatro_rotation(ps, pd, KT, rot_nb, p->rotation_gnc_factor, p->rotation_max_iterations,
                           p->rotation_cost_threshold, sol->rotation, rmask, &sol->gnc_cost,
                           &sol->gnc_iterations);
  else
    oracle_gnc_tls_rotation(ps, pd, KT, rot_nb, p->rotation_gnc_factor, p->rotation_max_iterations,
                            p->rotation_cost_threshold, sol->rotation, rmask, &sol->gnc_cost,
                            &sol->gnc_iterations);
  int nr = 0;
  for (int64_t i = 0; i < KT; ++i) /* :712-716 */
    if (rmask[i]) {
      if (rot_inliers_out && nr < rot_capacity) rot_inliers_out[nr] = (int32_t)i;
      ++nr;
    }
  sol->n_rotation_inliers = nr;
  free(rmask);
  free(ps);
  free(pd);

  /* :717-731 translation */
  double* a = (double*)malloc((size_t)(3 * K) * sizeof(double));
  double* b = (double*)malloc((size_t)(3 * K) * sizeof(double));
  const double* R = sol->rotation;
  for (int i = 0; i < K; ++i) {
    const double* s = src + 3 * (int64_t)clique[i];
    for (int r = 0; r < 3; ++r) {
      /* scale * rotation * src: Eigen evaluates (scale*R) then the product */
      double sr0 = sol->scale * R[3 * r], sr1 = sol->scale * R[3 * r + 1],
             sr2 = sol->scale * R[3 * r + 2];
      a[3 * i + r] = sr0 * s[0] + sr1 * s[1] + sr2 * s[2];
      b[3 * i + r] = dst[3 * (int64_t)clique[i] + r];
    }
  }
  uint8_t* tmask = (uint8_t*)malloc((size_t)K);
  oracle_tls_translation(a, b, K, p->noise_bound, p->cbar2, sol->translation, tmask);
  int nt = 0;
  for (int i = 0; i < K; ++i)
    if (tmask[i]) {
      if (trans_inliers_out) trans_inliers_out[nt] = i;
      ++nt;
    }
  sol->n_translation_inliers = nt;
  free(tmask);
  free(a);
  free(b);
  free(clique);
  sol->valid = 1; /* :734 */
  return 0;
}
