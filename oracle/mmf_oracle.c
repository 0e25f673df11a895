/*
 * mmf_oracle.c — CPU ORACLE for the hypergraph-construction hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under multimodal-fusion_amd/ may import, link or call this
 * file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the
 * checker / the reported CPU baseline.
 *
 * It restates, in plain C with the canonical arithmetic of include/mmf_hg.h, what the reference
 * (zz9tf/multimodal-fusion, paths relative to its root) computes on this path:
 *   - squared-L2 by norm expansion, (n_i + n_j) - 2*dot   build_hypergraph/similarity_kernel.py:43-49
 *   - Gaussian RBF exp(-lambda * sq)                      build_hypergraph/similarity_kernel.py:52, 84
 *   - product of the two kernels                          build_hypergraph/similarity_kernel.py:122
 *   - direct-difference RBF, one row against all          build_hypergraph/preprocess_hypergraph.py:254-256
 *   - Euclidean (k+1)-NN with self dropped                build_hypergraph/preprocess_hypergraph.py:379-388
 *   - per-edge max(0, cosine) weights, eps = 1e-8         build_hypergraph/preprocess_hypergraph.py:414-420
 *   - lower median of the off-diagonal + threshold scan   build_hypergraph/similarity_kernel.py:183-202
 *
 * Parity pinning: tests/test_oracle_golden.py checks every function here against golden vectors
 * produced by running the reference's own functions in the build container
 * (tests/golden/make_golden.py).  The reference's dot products come from BLAS (torch.mm) whose
 * summation order is unspecified, so scores are pinned to 1e-5 and indices are pinned exactly on
 * tie-free fixtures; the canonical chain below is what the HIP path must match bit for bit.
 *
 * Build: gcc -O3 -mavx2 -mfma -ffp-contract=off -fopenmp -shared -fPIC (oracle/build.py).
 * -ffp-contract=off is REQUIRED: every rounding below is part of the definition.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MMF_DOT 0
#define MMF_COSINE 1
#define MMF_NEG_SQ_L2 2
#define MMF_RBF 3
#define MMF_RBF_DIRECT 4

#define CB 16 /* columns per transposed panel */
#define RB 4  /* rows per register block      */

/* canonical k-ordered fmaf chain, include/mmf_hg.h */
static inline float chain(const float* a, const float* b, int64_t d) {
  float acc = 0.0f;
  for (int64_t k = 0; k < d; ++k) acc = __builtin_fmaf(a[k], b[k], acc);
  return acc;
}

static inline float sq_from(float ni, float nj, float dot) {
  float s = ni + nj;        /* similarity_kernel.py:49: (n_i + n_j) ... */
  float t = 2.0f * dot;     /* exact                                    */
  return s - t;             /*                       ... - 2*dot        */
}

static inline float cos_from(float ni, float nj, float dot) {
  float a = sqrtf(ni), b = sqrtf(nj);
  if (!(a > 1e-8f)) a = 1e-8f; /* F.cosine_similarity eps, preprocess_hypergraph.py:419 */
  if (!(b > 1e-8f)) b = 1e-8f;
  return dot / (a * b);
}

/* key = what is ranked, val = what is reported */
static inline void key_val(int metric, float lambda, float ni, float nj, float dot, float* key,
                           float* val) {
  switch (metric) {
    case MMF_DOT: *key = dot; *val = dot; break;
    case MMF_COSINE: *key = cos_from(ni, nj, dot); *val = *key; break;
    case MMF_NEG_SQ_L2: *key = -sq_from(ni, nj, dot); *val = *key; break;
    default: { /* MMF_RBF */
      float nl = -lambda;
      *key = nl * sq_from(ni, nj, dot);
      *val = expf(*key);
    }
  }
}

/* (key desc, id asc); NaN keys rank last */
static inline int better(float ka, int64_t ia, float kb, int64_t ib) {
  if (ka != ka) ka = -INFINITY;
  if (kb != kb) kb = -INFINITY;
  return ka > kb || (ka == kb && ia < ib);
}

typedef struct { float key, val; int64_t id; } ent_t;

static inline void topk_push(ent_t* t, int* cnt, int k, float key, float val, int64_t id) {
  int c = *cnt;
  if (c == k && !better(key, id, t[k - 1].key, t[k - 1].id)) return;
  int p = (c < k) ? c : k - 1;
  while (p > 0 && better(key, id, t[p - 1].key, t[p - 1].id)) { t[p] = t[p - 1]; --p; }
  t[p].key = key; t[p].val = val; t[p].id = id;
  if (c < k) *cnt = c + 1;
}

/*
 * Fused similarity + top-k (mmf_simtopk semantics).  X:[n,d], Y:[m,d] (NULL = X), f32.
 * Returns 0, or -1 on a bad argument (same conditions as the HIP path).
 */
int mmf_oracle_simtopk(const float* X, int64_t n, const float* Y, int64_t m, int64_t d, int metric,
                       float lambda, int k, int exclude_self, int64_t row_offset,
                       int64_t col_offset, int64_t* out_idx, float* out_val, int nthreads) {
  if (!Y) { Y = X; m = n; }
  if (n < 0 || m < 0 || d < 1 || k < 1 || metric < 0 || metric > MMF_RBF) return -1;
  if (metric == MMF_RBF && !(lambda > 0.0f)) return -1;
  if (n == 0) return 0;
  /* admissible columns per row: m, minus one when the row's own id falls in the column range */
  {
    int64_t lo = col_offset, hi = col_offset + m;
    for (int64_t i = 0; i < n; ++i) {
      int64_t g = row_offset + i;
      int64_t adm = m - ((exclude_self && g >= lo && g < hi) ? 1 : 0);
      if (k > adm) return -1;
    }
  }
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  float* nx = (float*)malloc(sizeof(float) * (size_t)n);
  float* ny = (float*)malloc(sizeof(float) * (size_t)m);
  int64_t np = (m + CB - 1) / CB;
  float* yt = (float*)aligned_alloc(64, sizeof(float) * (size_t)np * (size_t)d * CB);
  if (!nx || !ny || !yt) { free(nx); free(ny); free(yt); return -2; }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) nx[i] = chain(X + i * d, X + i * d, d);
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < m; ++j) ny[j] = chain(Y + j * d, Y + j * d, d);
  /* transposed panels: yt[p][k][c] = Y[p*CB + c][k] (zero for c past m) */
#pragma omp parallel for schedule(static)
  for (int64_t p = 0; p < np; ++p) {
    float* dst = yt + (size_t)p * d * CB;
    for (int64_t kk = 0; kk < d; ++kk)
      for (int c = 0; c < CB; ++c) {
        int64_t j = p * CB + c;
        dst[kk * CB + c] = (j < m) ? Y[j * d + kk] : 0.0f;
      }
  }
  int64_t nrb = (n + RB - 1) / RB;
#pragma omp parallel
  {
    ent_t* tk = (ent_t*)malloc(sizeof(ent_t) * (size_t)RB * (size_t)k);
    int cnt[RB];
#pragma omp for schedule(dynamic, 4)
    for (int64_t rb = 0; rb < nrb; ++rb) {
      int64_t i0 = rb * RB;
      int nr = (int)((n - i0 < RB) ? (n - i0) : RB);
      const float* xr[RB];
      for (int r = 0; r < RB; ++r) { xr[r] = X + (i0 + (r < nr ? r : 0)) * d; cnt[r] = 0; }
      for (int64_t p = 0; p < np; ++p) {
        const float* ytp = yt + (size_t)p * d * CB;
        float acc[RB][CB];
        for (int r = 0; r < RB; ++r)
          for (int c = 0; c < CB; ++c) acc[r][c] = 0.0f;
        for (int64_t kk = 0; kk < d; ++kk) {
          const float* yk = ytp + kk * CB;
          for (int r = 0; r < RB; ++r) {
            float xv = xr[r][kk];
            for (int c = 0; c < CB; ++c) acc[r][c] = __builtin_fmaf(xv, yk[c], acc[r][c]);
          }
        }
        for (int r = 0; r < nr; ++r) {
          int64_t gi = row_offset + i0 + r;
          for (int c = 0; c < CB; ++c) {
            int64_t j = p * CB + c;
            if (j >= m) break;
            int64_t gj = col_offset + j;
            if (exclude_self && gj == gi) continue;
            float key, val;
            key_val(metric, lambda, nx[i0 + r], ny[j], acc[r][c], &key, &val);
            topk_push(tk + (size_t)r * k, &cnt[r], k, key, val, gj);
          }
        }
      }
      for (int r = 0; r < nr; ++r)
        for (int t = 0; t < k; ++t) {
          out_idx[(i0 + r) * k + t] = tk[(size_t)r * k + t].id;
          out_val[(i0 + r) * k + t] = tk[(size_t)r * k + t].val;
        }
    }
    free(tk);
  }
  free(nx); free(ny); free(yt);
  return 0;
}

/* Dense [n,m] similarity (mmf_sim_dense semantics). */
int mmf_oracle_sim_dense(const float* X, int64_t n, const float* Y, int64_t m, int64_t d,
                         int metric, float lambda, float* out, int nthreads) {
  if (!Y) { Y = X; m = n; }
  if (n < 0 || m < 0 || d < 1 || metric < 0 || metric > MMF_RBF_DIRECT) return -1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  float* nx = (float*)malloc(sizeof(float) * (size_t)(n ? n : 1));
  float* ny = (float*)malloc(sizeof(float) * (size_t)(m ? m : 1));
  if (!nx || !ny) { free(nx); free(ny); return -2; }
  for (int64_t i = 0; i < n; ++i) nx[i] = chain(X + i * d, X + i * d, d);
  for (int64_t j = 0; j < m; ++j) ny[j] = chain(Y + j * d, Y + j * d, d);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j < m; ++j) {
      float v;
      if (metric == MMF_RBF_DIRECT) {
        /* preprocess_hypergraph.py:254-256: diff, square, sum, exp(-lambda_h * .) */
        float acc = 0.0f;
        for (int64_t kk = 0; kk < d; ++kk) {
          float t = X[i * d + kk] - Y[j * d + kk];
          acc = __builtin_fmaf(t, t, acc);
        }
        float nl = -lambda;
        v = expf(nl * acc);
      } else {
        float key;
        key_val(metric, lambda, nx[i], ny[j], chain(X + i * d, Y + j * d, d), &key, &v);
      }
      out[i * m + j] = v;
    }
  free(nx); free(ny);
  return 0;
}

/* K = K_h * K_g, similarity_kernel.py:116-122 */
int mmf_oracle_sim_dense_combined(const float* F, const float* P, int64_t n, int64_t d, int64_t dp,
                                  float lambda_h, float lambda_g, float* out, int nthreads) {
  if (n < 0 || d < 1 || dp < 1) return -1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  float* nf = (float*)malloc(sizeof(float) * (size_t)(n ? n : 1));
  float* npp = (float*)malloc(sizeof(float) * (size_t)(n ? n : 1));
  if (!nf || !npp) { free(nf); free(npp); return -2; }
  for (int64_t i = 0; i < n; ++i) {
    nf[i] = chain(F + i * d, F + i * d, d);
    npp[i] = chain(P + i * dp, P + i * dp, dp);
  }
  float nlh = -lambda_h, nlg = -lambda_g;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j < n; ++j) {
      float kh = expf(nlh * sq_from(nf[i], nf[j], chain(F + i * d, F + j * d, d)));
      float kg = expf(nlg * sq_from(npp[i], npp[j], chain(P + i * dp, P + j * dp, dp)));
      out[i * n + j] = kh * kg;
    }
  free(nf); free(npp);
  return 0;
}

/* w_e = max(0, cos(x_i, x_j)), preprocess_hypergraph.py:414-420.  edge_index:[2,E]. */
int mmf_oracle_edge_cosine(const float* X, int64_t n, int64_t d, const int64_t* edge_index,
                           int64_t E, float* out_w) {
  if (n < 0 || d < 1 || E < 0) return -1;
  for (int64_t e = 0; e < E; ++e) {
    int64_t i = edge_index[e], j = edge_index[E + e];
    if (i < 0 || i >= n || j < 0 || j >= n) return -1;
    const float *a = X + i * d, *b = X + j * d;
    float c = cos_from(chain(a, a, d), chain(b, b, d), chain(a, b, d));
    out_w[e] = (c > 0.0f) ? c : 0.0f; /* python max(0.0, w): NaN -> 0.0 */
  }
  return 0;
}

/* Merge two sorted [n,k] lists (mmf_topk_merge semantics). */
int mmf_oracle_topk_merge(const int64_t* ia, const float* va, const int64_t* ib, const float* vb,
                          int64_t n, int k, int64_t* io, float* vo) {
  if (n < 0 || k < 1) return -1;
  for (int64_t r = 0; r < n; ++r) {
    const int64_t *pa = ia + r * k, *pb = ib + r * k;
    const float *qa = va + r * k, *qb = vb + r * k;
    int a = 0, b = 0, o = 0;
    while (o < k) {
      while (a < k && pa[a] < 0) ++a;
      while (b < k && pb[b] < 0) ++b;
      int ta = a < k, tb = b < k;
      if (!ta && !tb) break;
      int pick_a;
      if (ta && tb) {
        if (pa[a] == pb[b]) { ++b; continue; } /* same id in both: keep once (from a) */
        pick_a = better(qa[a], pa[a], qb[b], pb[b]);
      } else {
        pick_a = ta;
      }
      int64_t id = pick_a ? pa[a] : pb[b];
      float v = pick_a ? qa[a] : qb[b];
      if (pick_a) ++a; else ++b;
      int dup = 0;
      for (int t = 0; t < o; ++t) if (io[r * k + t] == id) { dup = 1; break; }
      if (dup) continue;
      io[r * k + o] = id; vo[r * k + o] = v; ++o;
    }
    for (; o < k; ++o) { io[r * k + o] = -1; vo[r * k + o] = -INFINITY; }
  }
  return 0;
}

static int cmp_float(const void* a, const void* b) {
  float x = *(const float*)a, y = *(const float*)b;
  return (x > y) - (x < y);
}

/* torch.median over the n(n-1) off-diagonal entries: the LOWER middle element,
 * similarity_kernel.py:183-186. */
int mmf_oracle_offdiag_lower_median(const float* K, int64_t n, float* out_median) {
  if (n < 2) return -1;
  int64_t cnt = n * (n - 1);
  float* buf = (float*)malloc(sizeof(float) * (size_t)cnt);
  if (!buf) return -2;
  int64_t p = 0;
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j < n; ++j)
      if (i != j) buf[p++] = K[i * n + j];
  qsort(buf, (size_t)cnt, sizeof(float), cmp_float);
  *out_median = buf[(cnt - 1) / 2];
  free(buf);
  return 0;
}

/* Row-major threshold scan, self-loops kept; an entry is SKIPPED iff K < threshold
 * (similarity_kernel.py:193-202).  Writes at most `capacity` edges, returns the true count. */
int64_t mmf_oracle_threshold_edges(const float* K, int64_t n, float threshold, int64_t* edge_index,
                                   float* edge_w, int64_t capacity) {
  int64_t e = 0;
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j < n; ++j) {
      float s = K[i * n + j];
      if (s < threshold) continue;
      if (e < capacity) { edge_index[e] = i; edge_index[capacity + e] = j; edge_w[e] = s; }
      ++e;
    }
  return e;
}

int mmf_oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
