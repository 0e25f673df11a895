// mmf_kmeans.hip — the reference's KMeans(n_clusters, random_state=42, n_init=10).fit_predict on the device
// (build_hypergraph/preprocess_hypergraph.py:150-151, 299-300, 391-392), decision for decision.
//
// scikit-learn's fit is a sequence of DISCRETE decisions taken on float32 data: which point a uniform draw lands on in
// the cumulative closest-centre distances, which trial candidate has the smallest potential, which centre is nearest to
// a point, whether the labels changed, whether the centre shift is under the tolerance, which restart has the smallest
// inertia.  Its float32 sums go through BLAS (sdot / sgemv / sgemm) in a CPU- and thread-count-dependent order, so its
// own result is reproducible only up to decisions its rounding noise takes.  The contract here (restated on the CPU in
// oracle/kmeans_restate.py, which is what the tests check this file against, next to scikit-learn itself):
//
//   * every inner product and every sum is taken in FLOAT64 (v_mfma_f64_16x16x4_f64 for the distance contractions),
//     and rounded to float32 exactly where scikit-learn stores a float32: the mean-centred data, the seeding's
//     closest-centre distances float32(max(0, (-2 x.c + |c|^2) + |x|^2)), the potentials, the centres
//     float32(sum) * float32(1 / count);
//   * the caller supplies scikit-learn's random stream (numpy RandomState(seed), in scikit-learn's order of
//     consumption): the first centre of every restart and `trials` uniforms per seeding step;
//   * all n_init restarts advance in lockstep (their trajectories are independent: the stream is data-independent), each
//     with its own convergence state on the device; the host reads one small status block per Lloyd iteration.
//
// Bounds: the distance contractions (f64 MFMA, 78.6 TFLOP/s = 3.93e13 multiply-adds/s on MI355X); everything else is
// HBM-bound and small.  Algorithmic work per Lloyd iteration: n * n_init * k * d multiply-adds.
#include "mmf_dev.h"
#include "mmf_host.h"

namespace mmf {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int KM_T = 64;        // rows x points of a workgroup tile
constexpr int KM_KC = 32;       // k per staged chunk
constexpr int KM_LD = 34;       // f64 per LDS row: 68 dwords = 4 (mod 64) -> the 16 rows of a b128 fragment read hit 16 distinct bank quads
constexpr int KM_RUN = 0, KM_FINAL = 1, KM_DONE = 2, KM_RELOC = 3;

struct KmState {                // one per restart, device memory
  int state;                    // KM_RUN: E + M step; KM_FINAL: one more E step; KM_DONE; KM_RELOC: the host must relocate empty clusters
  int cur;                      // which of the two centre buffers holds the restart's current centres
  int changed;                  // set by the E step when a label differs from the previous iteration's
  int n_iter;
  int empty;                    // clusters without members found by the M step
  int pad;
  double shift;                 // sum over centres of |new - old|^2
  double inertia;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- column means, centring, tolerance, row norms ---------------------------------------------------------------
// colsum: partial[blk][col] = sum over the block's 256 rows of X[row][col] (pass 0) or of (X[row][col] - mean[col])^2
// (pass 1), f64, rows in ascending order per thread, four threads per column combined in a fixed order.
__global__ __launch_bounds__(256) void km_colsum_kernel(const float* __restrict__ X, int64_t n, int64_t d, const double* __restrict__ mean,
                                                        double* __restrict__ partial) {
  __shared__ double sh[4][64];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.y * 64 + c;
  const int64_t r0 = (int64_t)blockIdx.x * 256;
  double acc = 0.0;
  if (col < d) {
    const double mu = mean ? mean[col] : 0.0;
    for (int i = q; i < 256; i += 4) {
      const int64_t r = r0 + i;
      if (r < n) {
        const double v = (double)X[r * d + col] - mu;
        acc += mean ? v * v : v;
      }
    }
  }
  sh[q][c] = acc;
  __syncthreads();
  if (q == 0 && col < d) partial[(int64_t)blockIdx.x * d + col] = ((sh[0][c] + sh[1][c]) + sh[2][c]) + sh[3][c];
}

// one thread per column: mean (f64 and the float32 scikit-learn subtracts), or the variance; thread 0 of the last launch
// forms tol_abs = tol * mean(var).
__global__ void km_colfin_kernel(const double* __restrict__ partial, int64_t nblk, int64_t n, int64_t d, double* __restrict__ out64,
                                 float* __restrict__ out32) {
  const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= d) return;
  double s = 0.0;
  for (int64_t b = 0; b < nblk; ++b) s += partial[b * d + col];
  s /= (double)n;
  out64[col] = s;
  if (out32) out32[col] = (float)s;
}

// mean32[col] = (((x_0 + x_1) + x_2) + ... ) / float32(n): the float32 row-by-row accumulation numpy's X.mean(axis=0) performs
// on a C-contiguous float32 matrix, so the centred data are scikit-learn's bit for bit.  A sequential chain per column: one
// workgroup per 64 columns stages 64 x 64 tiles through LDS (all 256 threads load, the next tile's loads in flight) and its
// first wave walks the rows.
__global__ __launch_bounds__(256) void km_colmean_seq_kernel(const float* __restrict__ X, int64_t n, int64_t d, float* __restrict__ mean32) {
  __shared__ float tile[64][64];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * 64 + c;
  const bool live = col < d;
  float regs[16];
  float acc = 0.f;
  const int64_t nchunks = (n + 63) / 64;
  auto load = [&](int64_t ch) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t r = ch * 64 + q + 4 * i;
      regs[i] = (live && r < n) ? X[r * d + col] : 0.f;
    }
  };
  load(0);
  for (int64_t ch = 0; ch < nchunks; ++ch) {
#pragma unroll
    for (int i = 0; i < 16; ++i) tile[q + 4 * i][c] = regs[i];
    __syncthreads();
    if (ch + 1 < nchunks) load(ch + 1);
    if (q == 0) {
      const int64_t rows = (n - ch * 64 < 64) ? (n - ch * 64) : 64;
      if (rows == 64) {
#pragma unroll
        for (int r = 0; r < 64; ++r) acc += tile[r][c];
      } else {
        for (int r = 0; r < (int)rows; ++r) acc += tile[r][c];
      }
    }
    __syncthreads();
  }
  if (q == 0 && live) mean32[col] = acc / (float)n;
}

__global__ __launch_bounds__(1024) void km_tol_kernel(const double* __restrict__ var, int64_t d, double tol, double* __restrict__ tol_abs) {
  __shared__ double sh[1024];
  double s = 0.0;
  for (int64_t j = threadIdx.x; j < d; j += 1024) s += var[j];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *tol_abs = tol * (sh[0] / (double)d);
}

// Xc[row][j] = X[row][j] - mean32[j] (float32 subtraction: `X -= X_mean`), zero in the padding columns; one wave per
// row also leaves xx[row] = sum_j Xc[row][j]^2 in f64.
__global__ __launch_bounds__(256) void km_centre_kernel(const float* __restrict__ X, int64_t n, int64_t d, int64_t ds,
                                                        const float* __restrict__ mean32, float* __restrict__ Xc, double* __restrict__ xx) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  double acc = 0.0;
  for (int64_t j = lane; j < ds; j += 64) {
    float v = 0.f;
    if (j < d) v = X[row * d + j] - mean32[j];
    Xc[row * ds + j] = v;
    acc += (double)v * (double)v;
  }
  acc = wave_sum(acc);
  if (lane == 0) xx[row] = acc;
}

// cc[r] = |row r|^2 in f64 for the rows of a [R, ds] matrix (the centres of every restart), one wave per row.
__global__ __launch_bounds__(256) void km_rownorm_kernel(const float* __restrict__ C0, const float* __restrict__ C1, const KmState* __restrict__ st,
                                                         int64_t k, int64_t R, int64_t ds, double* __restrict__ cc) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int g = (int)(r / k);
  if (st[g].state == KM_DONE) return;
  const float* row = (st[g].cur ? C1 : C0) + r * ds;
  double acc = 0.0;
  for (int64_t j = lane; j < ds; j += 64) { const double v = (double)row[j]; acc += v * v; }
  acc = wave_sum(acc);
  if (lane == 0) cc[r] = acc;
}

// ---- the distance contraction: 64 rows x 64 points per workgroup on v_mfma_f64_16x16x4_f64 ---------------------------
// Four waves; wave w owns points 16 w .. 16 w + 15 against all 64 rows (four 16 x 16 accumulator tiles).  Row and point
// chunks of 32 k are converted to f64 on their way into LDS ([row][k], 34 f64 per row), double buffered, one barrier per
// chunk.  A lane of group g = lane >> 4 reads the f64 PAIR (8 q + 2 g, 8 q + 2 g + 1) of its row with one ds_read_b128
// and feeds the two to two consecutive MFMAs: inside a group of eight k the assignment of k to MFMA steps is a
// permutation shared by both operands, the sum over k is the same set of products.
// acc[m][v]: row 16 m + 4 v + (lane >> 4), point 16 w + (lane & 15)   (f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 v)
struct KmTile {
  double* As;   // [2][64][KM_LD]
  double* Bs;
};

__device__ __forceinline__ void km_stage_store(double* dst, const float4& v) {
  f64x2 a, b;
  a.x = (double)v.x; a.y = (double)v.y; b.x = (double)v.z; b.y = (double)v.w;
  *reinterpret_cast<f64x2*>(dst) = a;
  *reinterpret_cast<f64x2*>(dst + 2) = b;
}

// rowp[2] / ptp[2]: this thread's two staging rows (row t >> 3 and 32 + (t >> 3) of the tile) at column 4 (t & 7)
__device__ __forceinline__ void km_tile_dots(const float* const rowp[2], const float* const ptp[2], int64_t nchunks, double* As, double* Bs,
                                             f64x4 acc[4]) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, g = lane >> 4;
  const int sr = t >> 3, sk = (t & 7) * 4;
  float4 ra[2], rb[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    ra[u] = *reinterpret_cast<const float4*>(rowp[u]);
    rb[u] = *reinterpret_cast<const float4*>(ptp[u]);
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    km_stage_store(As + (sr + 32 * u) * KM_LD + sk, ra[u]);
    km_stage_store(Bs + (sr + 32 * u) * KM_LD + sk, rb[u]);
  }
  __syncthreads();
  for (int64_t c = 0; c < nchunks; ++c) {
    const int buf = (int)(c & 1);
    const bool more = c + 1 < nchunks;
    if (more) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        ra[u] = *reinterpret_cast<const float4*>(rowp[u] + (c + 1) * KM_KC);
        rb[u] = *reinterpret_cast<const float4*>(ptp[u] + (c + 1) * KM_KC);
      }
    }
    const double* A = As + buf * (KM_T * KM_LD) + (lane & 15) * KM_LD + 2 * g;
    const double* B = Bs + buf * (KM_T * KM_LD) + (16 * w + (lane & 15)) * KM_LD + 2 * g;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f64x2 b = *reinterpret_cast<const f64x2*>(B + 8 * q);
      f64x2 a[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) a[m] = *reinterpret_cast<const f64x2*>(A + m * 16 * KM_LD + 8 * q);
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].x, b.x, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].y, b.y, acc[m], 0, 0, 0);
    }
    if (more) {
      double* An = As + (buf ^ 1) * (KM_T * KM_LD);
      double* Bn = Bs + (buf ^ 1) * (KM_T * KM_LD);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        km_stage_store(An + (sr + 32 * u) * KM_LD + sk, ra[u]);
        km_stage_store(Bn + (sr + 32 * u) * KM_LD + sk, rb[u]);
      }
    }
    __syncthreads();
  }
}

// Seeding distances: out[r][i] = min(closest[r / group][i], float32(max(0, (-2 c_r.x_i + |c_r|^2) + |x_i|^2))) for the rows
// c_r = Xc[cand[r]], r < R, and partial[r][tile] = sum of out[r][i] over the tile's points (f64).  closest == NULL: no clamp.
__global__ __launch_bounds__(256, 2) void km_seed_dots_kernel(const float* __restrict__ Xc, int64_t n, int64_t ds, const double* __restrict__ xx,
                                                              const int64_t* __restrict__ cand, int64_t R, int64_t group,
                                                              const float* __restrict__ closest, float* __restrict__ out,
                                                              double* __restrict__ partial) {
  extern __shared__ double km_lds[];
  double* As = km_lds;
  double* Bs = km_lds + 2 * KM_T * KM_LD;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int64_t p0 = (int64_t)blockIdx.x * KM_T, r0 = (int64_t)blockIdx.y * KM_T;
  const float* rowp[2];
  const float* ptp[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    int64_t r = r0 + (t >> 3) + 32 * u;
    if (r >= R) r = R - 1;
    int64_t p = p0 + (t >> 3) + 32 * u;
    if (p >= n) p = n - 1;
    rowp[u] = Xc + cand[r] * ds + (t & 7) * 4;
    ptp[u] = Xc + p * ds + (t & 7) * 4;
  }
  f64x4 acc[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) acc[m] = f64x4{0.0, 0.0, 0.0, 0.0};
  km_tile_dots(rowp, ptp, ds / KM_KC, As, Bs, acc);
  // epilogue through LDS: [64 rows][65] f32
  float* T = reinterpret_cast<float*>(km_lds);
  {
    const int64_t p = p0 + 16 * w + (lane & 15);
    const double xp = xx[p < n ? p : n - 1];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int rl = 16 * m + 4 * v + (lane >> 4);
        int64_t r = r0 + rl;
        if (r >= R) r = R - 1;
        const double cr = xx[cand[r]];
        double dd = -2.0 * acc[m][v];
        dd = dd + cr;
        dd = dd + xp;
        float f = (float)dd;
        f = f > 0.f ? f : 0.f;
        T[rl * 65 + 16 * w + (lane & 15)] = f;
      }
  }
  __syncthreads();
  const int64_t p = p0 + lane;
  for (int rr = 0; rr < 16; ++rr) {
    const int rl = 16 * w + rr;
    const int64_t r = r0 + rl;
    if (r >= R) break;
    float f = T[rl * 65 + lane];
    double s = 0.0;
    if (p < n) {
      if (closest) { const float cl = closest[(r / group) * n + p]; f = f < cl ? f : cl; }
      out[r * n + p] = f;
      s = (double)f;
    }
    s = wave_sum(s);
    if (lane == 0) partial[r * gridDim.x + blockIdx.x] = s;
  }
}

// E step: label[g][i] = first arg-min over the k centres of restart g of |c|^2 - 2 x_i.c (f64); `changed[g]` when it
// differs from the label of the previous iteration.  Labels are stored combined (g k + centre): the segment ids of the M step.
__global__ __launch_bounds__(256, 2) void km_assign_kernel(const float* __restrict__ Xc, int64_t n, int64_t ds, const float* __restrict__ C0,
                                                           const float* __restrict__ C1, const double* __restrict__ cc, int64_t k,
                                                           KmState* __restrict__ st, int g0, int64_t* __restrict__ labels) {
  extern __shared__ double km_lds[];
  double* As = km_lds;
  double* Bs = km_lds + 2 * KM_T * KM_LD;
  const int g = g0 + blockIdx.y;
  if (st[g].state == KM_DONE) return;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int64_t p0 = (int64_t)blockIdx.x * KM_T;
  const float* C = (st[g].cur ? C1 : C0) + (int64_t)g * k * ds;
  const double* ccg = cc + (int64_t)g * k;
  const float* ptp[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    int64_t p = p0 + (t >> 3) + 32 * u;
    if (p >= n) p = n - 1;
    ptp[u] = Xc + p * ds + (t & 7) * 4;
  }
  double best = __builtin_huge_val();
  int bi = 0;
  for (int64_t r0 = 0; r0 < k; r0 += KM_T) {
    const float* rowp[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int64_t r = r0 + (t >> 3) + 32 * u;
      if (r >= k) r = k - 1;
      rowp[u] = C + r * ds + (t & 7) * 4;
    }
    f64x4 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = f64x4{0.0, 0.0, 0.0, 0.0};
    km_tile_dots(rowp, ptp, ds / KM_KC, As, Bs, acc);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int64_t r = r0 + 16 * m + 4 * v + (lane >> 4);     // ascending in (m, v) for a lane: strict < keeps the first minimum
        if (r < k) {
          const double s = ccg[r] - 2.0 * acc[m][v];
          if (s < best) { best = s; bi = (int)r; }
        }
      }
  }
#pragma unroll
  for (int o = 16; o <= 32; o <<= 1) {
    const double ob = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  const int64_t p = p0 + 16 * w + (lane & 15);
  if ((lane >> 4) == 0 && p < n) {
    const int64_t lab = (int64_t)g * k + bi;
    int64_t* dst = labels + (int64_t)g * n + p;
    if (*dst != lab) { *dst = lab; st[g].changed = 1; }
  }
}

// ---- seeding control -----------------------------------------------------------------------------------------------
__global__ void km_seed_first_kernel(const int64_t* __restrict__ first, int64_t n_init, int64_t k, int64_t* __restrict__ seeds,
                                     int64_t* __restrict__ cand) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_init) return;
  seeds[i * k] = first[i];
  cand[i] = first[i];
}

// One workgroup per restart: candidate of trial t = searchsorted(cumsum(closest as f64), u_t * float64(pot32), side='left'),
// clipped to n - 1.  A thread owns a run of consecutive points; the 1024 run totals are scanned in LDS; a target lands in a run
// by binary search and on a point by walking the run.  amb[0] counts the draws that fall within 4 float32 ulps of the potential
// of a boundary of the cumulative sum: decisions that scikit-learn's own float32 potential (a BLAS sum) may take either way.
__global__ __launch_bounds__(1024) void km_seed_draw_kernel(const float* __restrict__ closest, int64_t n, const float* __restrict__ pot32,
                                                            const double* __restrict__ U, int64_t u_stride, int trials,
                                                            int64_t* __restrict__ cand, unsigned int* __restrict__ amb) {
  __shared__ double part[1024];
  const int t = threadIdx.x;
  const float* cl = closest + (int64_t)blockIdx.x * n;
  const int64_t per = (n + 1023) / 1024, b = (int64_t)t * per;
  int64_t e = b + per;
  if (e > n) e = n;
  double sum = 0.0;
  for (int64_t j = b; j < e; ++j) sum += (double)cl[j];
  part[t] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const double v = (t >= o) ? part[t - o] : 0.0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  if (t < trials) {
    const double pot = (double)pot32[blockIdx.x];
    const double target = U[(int64_t)blockIdx.x * u_stride + t] * pot;
    int lo = 0, hi = 1023;                      // first run whose inclusive total reaches the target
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (part[mid] >= target) hi = mid; else lo = mid + 1;
    }
    double run = lo ? part[lo - 1] : 0.0;
    int64_t j = (int64_t)lo * per, last = j + per;
    if (last > n) last = n;
    int64_t pick = n - 1;
    double below = run, at = run;
    for (; j < last; ++j) {
      below = run;
      run += (double)cl[j];
      at = run;
      if (run >= target) { pick = j; break; }
    }
    if (pick > n - 1) pick = n - 1;
    if (pick < 0) pick = 0;
    const double band = 4.0 * 1.1920928955078125e-07 * pot;
    if (target - below <= band || at - target <= band) atomicAdd(amb, 1u);
    cand[(int64_t)blockIdx.x * trials + t] = pick;
  }
}

// One restart per blockIdx.x (the row copy shared by gridDim.y workgroups): potential of every trial = float32 of the f64 sum
// of its tile partials; the first smallest wins (np.argmin over float32 potentials); its row becomes the restart's closest-centre
// distances.  rows == closest (the first centre): nothing to copy.
__global__ __launch_bounds__(256) void km_seed_choose_kernel(const float* __restrict__ rows, const double* __restrict__ partial, int64_t ntile,
                                                             int64_t n, int trials, const int64_t* __restrict__ cand,
                                                             float* __restrict__ closest, float* __restrict__ pot32,
                                                             int64_t* __restrict__ seeds, int64_t k, int64_t step, unsigned int* __restrict__ amb) {
  __shared__ double red[256];
  __shared__ float pots[64];
  __shared__ int best_s;
  const int t = threadIdx.x;
  const int64_t i = blockIdx.x;
  for (int tr = 0; tr < trials; ++tr) {
    const double* p = partial + (i * trials + tr) * ntile;
    double sum = 0.0;
    for (int64_t q = t; q < ntile; q += 256) sum += p[q];
    red[t] = sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (t < o) red[t] += red[t + o];
      __syncthreads();
    }
    if (t == 0) pots[tr] = (float)red[0];
    __syncthreads();
  }
  if (t == 0) {
    int bsel = 0;
    for (int tr = 1; tr < trials; ++tr)
      if (pots[tr] < pots[bsel]) bsel = tr;
    best_s = bsel;
    if (blockIdx.y == 0) {     // another candidate within 4 float32 ulps of the winner: scikit-learn's BLAS sums may order them either way
      bool close = false;
      for (int tr = 0; tr < trials; ++tr)
        if (tr != bsel && cand[i * trials + tr] != cand[i * trials + bsel] && pots[tr] - pots[bsel] <= 4.f * 1.1920928955078125e-07f * pots[bsel]) close = true;
      if (close) atomicAdd(amb + 1, 1u);
    }
  }
  __syncthreads();
  const int bs = best_s;
  const float* src = rows + (i * trials + bs) * n;
  float* dst = closest + i * n;
  if (src != dst)
    for (int64_t j = (int64_t)blockIdx.y * 256 + t; j < n; j += (int64_t)gridDim.y * 256) dst[j] = src[j];
  if (t == 0 && blockIdx.y == 0) {
    seeds[i * k + step] = cand[i * trials + bs];
    pot32[i] = pots[bs];
  }
}

// C[0][g][r][:] = Xc[seeds[g][r]][:]
__global__ __launch_bounds__(256) void km_gather_kernel(const float* __restrict__ Xc, int64_t ds, const int64_t* __restrict__ seeds, int64_t R,
                                                        float* __restrict__ C) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* src = Xc + seeds[r] * ds;
  for (int64_t j = threadIdx.x & 63; j < ds; j += 64) C[r * ds + j] = src[j];
}

// ---- M step ------------------------------------------------------------------------------------------------------------
// Workgroup = (cluster of a restart, strip of 64 columns), 4 waves; wave w adds the members w, w + 4, ... in member order in
// f64, the four partial sums are combined in wave order.  centre = float32(sum) * float32(1 / count) (scikit-learn's
// _average_centers; an empty cluster keeps its zero sum).  sums32 keeps float32(sum) for the relocation of empty clusters;
// shift_part[seg][strip] = sum over the strip of (new - old)^2.
__global__ __launch_bounds__(256) void km_update_kernel(const float* __restrict__ Xc, int64_t n, int64_t ds, const int64_t* __restrict__ order,
                                                        const int64_t* __restrict__ offsets, int64_t k, float* __restrict__ C0,
                                                        float* __restrict__ C1, float* __restrict__ sums32, KmState* __restrict__ st, int g0,
                                                        double* __restrict__ shift_part) {
  __shared__ double part[4][64];
  const int64_t seg = (int64_t)g0 * k + blockIdx.x;
  const int g = (int)(seg / k);
  if (st[g].state != KM_RUN) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.y * 64 + lane;
  const bool live = col < ds;
  const int64_t b = offsets[seg], e = offsets[seg + 1];
  const int64_t base = (int64_t)g * n;
  double acc = 0.0;
  if (live) {
    int64_t q = b + w;
    for (; q + 12 < e; q += 16) {
      const int64_t r0 = order[q] - base, r1 = order[q + 4] - base, r2 = order[q + 8] - base, r3 = order[q + 12] - base;
      const float x0 = Xc[r0 * ds + col], x1 = Xc[r1 * ds + col], x2 = Xc[r2 * ds + col], x3 = Xc[r3 * ds + col];
      acc += (double)x0; acc += (double)x1; acc += (double)x2; acc += (double)x3;
    }
    for (; q < e; q += 4) acc += (double)Xc[(order[q] - base) * ds + col];
  }
  part[w][lane] = acc;
  __syncthreads();
  if (w == 0) {
    const double sum = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
    const float s32 = (float)sum;
    const int64_t cnt = e - b;
    float cn = s32;
    if (cnt > 0) { const float alpha = (float)(1.0 / (double)(float)cnt); cn = s32 * alpha; }
    const int cur = st[g].cur;
    const float* Co = cur ? C1 : C0;
    float* Cn = cur ? C0 : C1;
    double dlt = 0.0;
    if (live) {
      sums32[seg * ds + col] = s32;
      dlt = (double)cn - (double)Co[seg * ds + col];
      Cn[seg * ds + col] = cn;
    }
    const double sh = wave_sum(dlt * dlt);
    if (lane == 0) {
      shift_part[seg * gridDim.y + blockIdx.y] = sh;
      if (cnt == 0 && blockIdx.y == 0) atomicAdd(&st[g].empty, 1);
    }
  }
}

// One thread per restart, after the E (+ M) step of Lloyd iteration `it`: scikit-learn's convergence logic
// (_kmeans_single_lloyd): swap the centre buffers; unchanged labels -> done, no further E step; shift <= tol or the last
// iteration -> one more E step with the new centres.  A restart with empty clusters waits for the host (KM_RELOC).
__global__ void km_state_kernel(KmState* __restrict__ st, int n_init, int64_t k, int64_t strips, const double* __restrict__ shift_part,
                                const double* __restrict__ tol_abs, int it, int max_iter, int only, int* __restrict__ status) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_init) return;
  if (only < 0 || only == g) {
    KmState s = st[g];
    if (s.state == KM_FINAL) {
      s.state = KM_DONE;
    } else if (s.state == KM_RUN || s.state == KM_RELOC) {
      if (s.state == KM_RUN && s.empty > 0) {
        s.state = KM_RELOC;
      } else {
        double sh = 0.0;
        for (int64_t q = 0; q < k * strips; ++q) sh += shift_part[(int64_t)g * k * strips + q];
        s.shift = sh;
        s.cur ^= 1;
        s.n_iter = it + 1;
        s.empty = 0;
        if (!s.changed) s.state = KM_DONE;
        else if (sh <= *tol_abs || it + 1 >= max_iter) s.state = KM_FINAL;
        else s.state = KM_RUN;
        s.changed = 0;
      }
    }
    st[g] = s;
  }
  status[g] = st[g].state;
}

// Relocation of empty clusters (scikit-learn's _relocate_empty_clusters_dense), one workgroup per restart that needs it:
// the points farthest from their (old) centre, in descending order (ties: lowest index), become the centres of the empty
// clusters in ascending cluster order; each is removed from its donor's float32 sum.  Then all k centres of the restart are
// re-averaged and the shift recomputed.  dist: [n] f64 scratch.
__global__ __launch_bounds__(1024) void km_relocate_kernel(const float* __restrict__ Xc, int64_t n, int64_t ds, int64_t k, int g,
                                                           const int64_t* __restrict__ labels, const int64_t* __restrict__ offsets,
                                                           float* __restrict__ C0, float* __restrict__ C1, float* __restrict__ sums32,
                                                           const KmState* __restrict__ st, double* __restrict__ dist, float* __restrict__ cnt,
                                                           double* __restrict__ shift_part, int64_t strips) {
  __shared__ double rv[1024];
  __shared__ int64_t ri[1024];
  __shared__ int64_t far_s;
  const int t = threadIdx.x;
  const int cur = st[g].cur;
  const float* Co = (cur ? C1 : C0) + (int64_t)g * k * ds;
  float* Cn = (cur ? C0 : C1) + (int64_t)g * k * ds;
  float* S = sums32 + (int64_t)g * k * ds;
  const int64_t* lab = labels + (int64_t)g * n;
  for (int64_t c = t; c < k; c += 1024) cnt[c] = (float)(offsets[(int64_t)g * k + c + 1] - offsets[(int64_t)g * k + c]);
  double mx = 0.0;
  for (int64_t i = t; i < n; i += 1024) {
    const float* c = Co + (lab[i] - (int64_t)g * k) * ds;
    double a = 0.0;
    for (int64_t j = 0; j < ds; ++j) { const double v = (double)Xc[i * ds + j] - (double)c[j]; a += v * v; }
    dist[i] = a;
    mx = a > mx ? a : mx;
  }
  rv[t] = mx;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (t < o) rv[t] = rv[t] > rv[t + o] ? rv[t] : rv[t + o];
    __syncthreads();
  }
  const bool pointless = !(rv[0] > 0.0);      // more clusters than distinct points: scikit-learn leaves the empty centres at zero
  __syncthreads();
  if (!pointless) {
    for (int64_t c = 0; c < k; ++c) {
      if (offsets[(int64_t)g * k + c + 1] != offsets[(int64_t)g * k + c]) continue;   // only the clusters that were empty BEFORE the relocation
      double bv = -1.0;
      int64_t bidx = n;
      for (int64_t i = t; i < n; i += 1024) {
        const double v = dist[i];
        if (v > bv) { bv = v; bidx = i; }
      }
      rv[t] = bv; ri[t] = bidx;
      __syncthreads();
      for (int o = 512; o > 0; o >>= 1) {
        if (t < o) {
          if (rv[t + o] > rv[t] || (rv[t + o] == rv[t] && ri[t + o] < ri[t])) { rv[t] = rv[t + o]; ri[t] = ri[t + o]; }
        }
        __syncthreads();
      }
      if (t == 0) { far_s = ri[0]; dist[ri[0]] = -2.0; }
      __syncthreads();
      const int64_t f = far_s, old = lab[f] - (int64_t)g * k;
      for (int64_t j = t; j < ds; j += 1024) {
        const float x = Xc[f * ds + j];
        S[old * ds + j] -= x;
        S[c * ds + j] = x;
      }
      __syncthreads();
      if (t == 0) { cnt[c] = 1.f; cnt[old] -= 1.f; }
      __syncthreads();
    }
  }
  // re-average and recompute the shift of the whole restart (strip partials: everything into strip 0 of each centre)
  for (int64_t c = 0; c < k; ++c) {
    const float wgt = cnt[c];
    double a = 0.0;
    for (int64_t j = t; j < ds; j += 1024) {
      float v = S[c * ds + j];
      if (wgt > 0.f) { const float alpha = (float)(1.0 / (double)wgt); v = v * alpha; }
      Cn[c * ds + j] = v;
      const double dl = (double)v - (double)Co[c * ds + j];
      a += dl * dl;
    }
    rv[t] = a;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
      if (t < o) rv[t] += rv[t + o];
      __syncthreads();
    }
    if (t == 0) {
      for (int64_t q = 0; q < strips; ++q) shift_part[((int64_t)g * k + c) * strips + q] = 0.0;
      shift_part[((int64_t)g * k + c) * strips] = rv[0];
    }
    __syncthreads();
  }
}

// ---- inertia, best restart, outputs -------------------------------------------------------------------------------------
// partial[g][blk] = sum over the block's points of |x_i - c_label|^2 (f64), one wave per point.
__global__ __launch_bounds__(256) void km_inertia_kernel(const float* __restrict__ Xc, int64_t n, int64_t ds, const float* __restrict__ C0,
                                                         const float* __restrict__ C1, const KmState* __restrict__ st,
                                                         const int64_t* __restrict__ labels, double* __restrict__ partial) {
  __shared__ double sh[4];
  const int g = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* C = st[g].cur ? C1 : C0;
  double acc = 0.0;
  for (int u = 0; u < 16; ++u) {
    const int64_t i = (int64_t)blockIdx.x * 64 + u * 4 + w;
    if (i >= n) break;
    const float* c = C + labels[(int64_t)g * n + i] * ds;
    const float* x = Xc + i * ds;
    double a = 0.0;
    for (int64_t j = lane; j < ds; j += 64) { const double v = (double)x[j] - (double)c[j]; a += v * v; }
    acc += wave_sum(a);
  }
  if (lane == 0) sh[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[(int64_t)g * gridDim.x + blockIdx.x] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

// Single workgroup: inertia of every restart, then scikit-learn's choice (KMeans.fit :1525-1532): a later restart replaces the
// best one when its inertia is strictly smaller AND its labels are not a function of the best's (_is_same_clustering).
// map: [k] int scratch.
__global__ __launch_bounds__(1024) void km_pick_kernel(KmState* __restrict__ st, int n_init, int64_t n, int64_t k, const double* __restrict__ partial,
                                                       int64_t nblk, const int64_t* __restrict__ labels, int* __restrict__ map, int* __restrict__ best_out) {
  __shared__ double red[1024];
  __shared__ int differs;
  const int t = threadIdx.x;
  for (int g = 0; g < n_init; ++g) {
    double s = 0.0;
    for (int64_t q = t; q < nblk; q += 1024) s += partial[(int64_t)g * nblk + q];
    red[t] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
      if (t < o) red[t] += red[t + o];
      __syncthreads();
    }
    if (t == 0) st[g].inertia = red[0];
    __syncthreads();
  }
  int best = 0;
  for (int g = 1; g < n_init; ++g) {
    if (!(st[g].inertia < st[best].inertia)) continue;
    // labels of g a function of labels... scikit-learn maps labels1 = the NEW restart's labels to labels2 = the best's
    for (int64_t c = t; c < k; c += 1024) map[c] = -1;
    if (t == 0) differs = 0;
    __syncthreads();
    for (int64_t i = t; i < n; i += 1024) {
      const int a = (int)(labels[(int64_t)g * n + i] - (int64_t)g * k), b = (int)(labels[(int64_t)best * n + i] - (int64_t)best * k);
      const int prev = atomicCAS(&map[a], -1, b);
      if (prev != -1 && prev != b) differs = 1;
    }
    __syncthreads();
    if (differs) best = g;
    __syncthreads();
  }
  if (t == 0) *best_out = best;
}

__global__ __launch_bounds__(256) void km_finish_kernel(const KmState* __restrict__ st, const int* __restrict__ best_p, int64_t n, int64_t k, int64_t d,
                                                        int64_t ds, const int64_t* __restrict__ labels, const float* __restrict__ C0,
                                                        const float* __restrict__ C1, const float* __restrict__ mean32,
                                                        int64_t* __restrict__ out_labels, float* __restrict__ out_centres) {
  const int best = *best_p;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out_labels[i] = labels[(int64_t)best * n + i] - (int64_t)best * k;
  if (out_centres && i < k * d) {
    const int64_t r = i / d, j = i % d;
    const float* C = (st[best].cur ? C1 : C0) + ((int64_t)best * k + r) * ds;
    out_centres[i] = C[j] + mean32[j];
  }
}

__global__ void km_init_state_kernel(KmState* st, int n_init) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_init) return;
  KmState s;
  s.state = KM_RUN; s.cur = 0; s.changed = 0; s.n_iter = 0; s.empty = 0; s.pad = 0; s.shift = 0.0; s.inertia = 0.0;
  st[g] = s;
}

// ---- host ----------------------------------------------------------------------------------------------------------------
static inline size_t al(size_t b) { return (b + 255) & ~size_t(255); }
static int64_t km_ds(int64_t d) { return (d + KM_KC - 1) / KM_KC * KM_KC; }

size_t kmeans_scratch_bytes(int64_t n, int64_t d, int64_t k, int64_t n_init, int trials) {
  const int64_t ds = km_ds(d), R = n_init * trials, ntile = (n + KM_T - 1) / KM_T, nblk = (n + 255) / 256, S = n_init * k;
  const int64_t strips = (ds + 63) / 64;
  size_t b = 0;
  b += al((size_t)n * ds * 4);                 // Xc
  b += al((size_t)n * 8);                      // xx
  b += al((size_t)nblk * d * 8);               // column partials
  b += 3 * al((size_t)d * 8) + al((size_t)d * 4);   // mean64, var64, (spare), mean32
  b += al(256);                                // tol_abs, best, amb
  b += al((size_t)R * n * 4);                  // trial rows
  b += al((size_t)n_init * n * 4);             // closest
  b += al((size_t)R * ntile * 8);              // row partials
  b += al((size_t)R * 8) + al((size_t)n_init * 8) + al((size_t)n_init * 4);   // cand, first, pot32
  b += al((size_t)n_init * (k > 1 ? k - 1 : 1) * trials * 8);                 // uniforms
  b += al((size_t)S * 8);                      // seeds
  b += 2 * al((size_t)S * ds * 4);             // centres x 2
  b += al((size_t)S * ds * 4);                 // sums32
  b += al((size_t)S * 8);                      // cc
  b += al((size_t)n_init * n * 8) * 2;         // labels, order
  b += al((size_t)S * 8) + al((size_t)(S + 1) * 8);   // counts, offsets
  b += al(segment_sort_scratch_bytes(n_init * n, S)) + al(4);
  b += al((size_t)S * strips * 8);             // shift partials
  b += al((size_t)n_init * sizeof(KmState)) + al((size_t)n_init * 4);
  b += al((size_t)n * 8) + al((size_t)k * 4);  // relocation scratch
  b += al((size_t)n_init * ntile * 8);         // inertia partials
  b += al((size_t)k * 4);                      // map
  return b + 4096;
}

int launch_kmeans_fit(const float* X, int64_t n, int64_t d, int64_t k, int64_t n_init, int trials, const int64_t* first_h,
                      const double* u_h, int max_iter, double tol, int64_t* out_labels, float* out_centres, int64_t* out_seeds,
                      double* info_h, void* scratch, hipStream_t s) {
  const int64_t ds = km_ds(d), R = n_init * trials, ntile = (n + KM_T - 1) / KM_T, nblk = (n + 255) / 256, S = n_init * k;
  const int64_t strips = (ds + 63) / 64;
  char* p = static_cast<char*>(scratch);
  auto take = [&](size_t bytes) { char* q = p; p += al(bytes); return q; };
  float* Xc = (float*)take((size_t)n * ds * 4);
  double* xx = (double*)take((size_t)n * 8);
  double* colpart = (double*)take((size_t)nblk * d * 8);
  double* mean64 = (double*)take((size_t)d * 8);
  double* var64 = (double*)take((size_t)d * 8);
  (void)take((size_t)d * 8);
  float* mean32 = (float*)take((size_t)d * 4);
  char* misc = take(256);
  double* tol_abs = (double*)misc;
  int* best = (int*)(misc + 8);
  unsigned int* amb = (unsigned int*)(misc + 16);
  float* rows = (float*)take((size_t)R * n * 4);
  float* closest = (float*)take((size_t)n_init * n * 4);
  double* rpart = (double*)take((size_t)R * ntile * 8);
  int64_t* cand = (int64_t*)take((size_t)R * 8);
  int64_t* first = (int64_t*)take((size_t)n_init * 8);
  float* pot32 = (float*)take((size_t)n_init * 4);
  const size_t u_count = (size_t)n_init * (k > 1 ? k - 1 : 1) * trials;
  double* U = (double*)take(u_count * 8);
  int64_t* seeds = (int64_t*)take((size_t)S * 8);
  float* C0 = (float*)take((size_t)S * ds * 4);
  float* C1 = (float*)take((size_t)S * ds * 4);
  float* sums32 = (float*)take((size_t)S * ds * 4);
  double* cc = (double*)take((size_t)S * 8);
  int64_t* labels = (int64_t*)take((size_t)n_init * n * 8);
  int64_t* order = (int64_t*)take((size_t)n_init * n * 8);
  int64_t* counts = (int64_t*)take((size_t)S * 8);
  int64_t* offsets = (int64_t*)take((size_t)(S + 1) * 8);
  void* sort_scratch = take(segment_sort_scratch_bytes(n_init * n, S));
  uint32_t* bad = (uint32_t*)take(4);
  double* shift_part = (double*)take((size_t)S * strips * 8);
  KmState* st = (KmState*)take((size_t)n_init * sizeof(KmState));
  int* status = (int*)take((size_t)n_init * 4);
  double* dist = (double*)take((size_t)n * 8);
  float* cntf = (float*)take((size_t)k * 4);
  double* ipart = (double*)take((size_t)n_init * ntile * 8);
  int* map = (int*)take((size_t)k * 4);

  const size_t lds = (size_t)4 * KM_T * KM_LD * 8;
  MMF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(km_seed_dots_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  MMF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(km_assign_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));

  // the caller's stream of random numbers
  MMF_HIP(hipMemcpyAsync(first, first_h, (size_t)n_init * 8, hipMemcpyHostToDevice, s));
  if (k > 1) MMF_HIP(hipMemcpyAsync(U, u_h, u_count * 8, hipMemcpyHostToDevice, s));
  MMF_HIP(hipMemsetAsync(misc, 0, 256, s));

  // centring, tolerance, norms
  const dim3 cgrid((unsigned)nblk, (unsigned)((d + 63) / 64));
  hipLaunchKernelGGL(km_colsum_kernel, cgrid, dim3(256), 0, s, X, n, d, (const double*)nullptr, colpart);
  MMF_LAUNCH_CHECK();
  hipLaunchKernelGGL(km_colfin_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, s, colpart, nblk, n, d, mean64, (float*)nullptr);
  MMF_LAUNCH_CHECK();
  hipLaunchKernelGGL(km_colmean_seq_kernel, dim3((unsigned)((d + 63) / 64)), dim3(256), 0, s, X, n, d, mean32);
  MMF_LAUNCH_CHECK();
  hipLaunchKernelGGL(km_colsum_kernel, cgrid, dim3(256), 0, s, X, n, d, (const double*)mean64, colpart);
  MMF_LAUNCH_CHECK();
  hipLaunchKernelGGL(km_colfin_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, s, colpart, nblk, n, d, var64, (float*)nullptr);
  MMF_LAUNCH_CHECK();
  hipLaunchKernelGGL(km_tol_kernel, dim3(1), dim3(1024), 0, s, var64, d, tol, tol_abs);
  MMF_LAUNCH_CHECK();
  hipLaunchKernelGGL(km_centre_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, X, n, d, ds, mean32, Xc, xx);
  MMF_LAUNCH_CHECK();

  // k-means++ for all restarts in lockstep
  hipLaunchKernelGGL(km_seed_first_kernel, dim3((unsigned)((n_init + 63) / 64)), dim3(64), 0, s, first, n_init, k, seeds, cand);
  MMF_LAUNCH_CHECK();
  unsigned choose_split = (unsigned)((n + 2047) / 2048);
  if (choose_split > 32) choose_split = 32;
  if (choose_split < 1) choose_split = 1;
  hipLaunchKernelGGL(km_seed_dots_kernel, dim3((unsigned)ntile, (unsigned)((n_init + KM_T - 1) / KM_T)), dim3(256), lds, s, Xc, n, ds, xx, cand, n_init,
                     (int64_t)1, (const float*)nullptr, closest, rpart);
  MMF_LAUNCH_CHECK();
  hipLaunchKernelGGL(km_seed_choose_kernel, dim3((unsigned)n_init, 1), dim3(256), 0, s, closest, rpart, ntile, n, 1, cand, closest, pot32, seeds, k,
                     (int64_t)0, amb);
  MMF_LAUNCH_CHECK();
  for (int64_t step = 1; step < k; ++step) {
    hipLaunchKernelGGL(km_seed_draw_kernel, dim3((unsigned)n_init), dim3(1024), 0, s, closest, n, pot32, U + (step - 1) * trials, (k - 1) * trials, trials,
                       cand, amb);
    MMF_LAUNCH_CHECK();
    hipLaunchKernelGGL(km_seed_dots_kernel, dim3((unsigned)ntile, (unsigned)((R + KM_T - 1) / KM_T)), dim3(256), lds, s, Xc, n, ds, xx, cand, R,
                       (int64_t)trials, closest, rows, rpart);
    MMF_LAUNCH_CHECK();
    hipLaunchKernelGGL(km_seed_choose_kernel, dim3((unsigned)n_init, choose_split), dim3(256), 0, s, rows, rpart, ntile, n, trials, cand, closest, pot32,
                       seeds, k, step, amb);
    MMF_LAUNCH_CHECK();
  }
  if (out_seeds) MMF_HIP(hipMemcpyAsync(out_seeds, seeds, (size_t)S * 8, hipMemcpyDeviceToDevice, s));

  // Lloyd iterations, all restarts in lockstep
  hipLaunchKernelGGL(km_gather_kernel, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, s, Xc, ds, seeds, S, C0);
  MMF_LAUNCH_CHECK();
  hipLaunchKernelGGL(km_init_state_kernel, dim3((unsigned)((n_init + 63) / 64)), dim3(64), 0, s, st, (int)n_init);
  MMF_LAUNCH_CHECK();
  MMF_HIP(hipMemsetAsync(labels, 0xff, (size_t)n_init * n * 8, s));      // -1: every label "changes" in the first iteration
  std::vector<int> h_status((size_t)n_init, KM_RUN);
  int it = 0;
  for (;; ++it) {
    bool any_run = false, any_live = false;
    for (int64_t g = 0; g < n_init; ++g) { any_run |= h_status[g] == KM_RUN; any_live |= h_status[g] != KM_DONE; }
    if (!any_live) break;
    hipLaunchKernelGGL(km_rownorm_kernel, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, s, C0, C1, st, k, S, ds, cc);
    MMF_LAUNCH_CHECK();
    hipLaunchKernelGGL(km_assign_kernel, dim3((unsigned)ntile, (unsigned)n_init), dim3(256), lds, s, Xc, n, ds, C0, C1, cc, k, st, 0, labels);
    MMF_LAUNCH_CHECK();
    if (any_run) {
      MMF_TRY(launch_segment_sort(labels, n_init * n, S, counts, offsets, order, sort_scratch, bad, s));
      hipLaunchKernelGGL(km_update_kernel, dim3((unsigned)S, (unsigned)strips), dim3(256), 0, s, Xc, n, ds, order, offsets, k, C0, C1, sums32, st, 0,
                         shift_part);
      MMF_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(km_state_kernel, dim3((unsigned)((n_init + 63) / 64)), dim3(64), 0, s, st, (int)n_init, k, strips, shift_part, tol_abs, it, max_iter,
                       -1, status);
    MMF_LAUNCH_CHECK();
    MMF_HIP(hipMemcpyAsync(h_status.data(), status, (size_t)n_init * 4, hipMemcpyDeviceToHost, s));
    MMF_HIP(hipStreamSynchronize(s));
    for (int64_t g = 0; g < n_init; ++g) {
      if (h_status[g] != KM_RELOC) continue;
      hipLaunchKernelGGL(km_relocate_kernel, dim3(1), dim3(1024), 0, s, Xc, n, ds, k, (int)g, labels, offsets, C0, C1, sums32, st, dist, cntf, shift_part,
                         strips);
      MMF_LAUNCH_CHECK();
      hipLaunchKernelGGL(km_state_kernel, dim3((unsigned)((n_init + 63) / 64)), dim3(64), 0, s, st, (int)n_init, k, strips, shift_part, tol_abs, it, max_iter,
                         (int)g, status);
      MMF_LAUNCH_CHECK();
      MMF_HIP(hipMemcpyAsync(h_status.data(), status, (size_t)n_init * 4, hipMemcpyDeviceToHost, s));
      MMF_HIP(hipStreamSynchronize(s));
    }
    if (it > max_iter + 2) { set_error("kmeans_fit: the convergence state machine did not terminate (internal invariant)"); return MMF_E_INTERNAL; }
  }

  hipLaunchKernelGGL(km_inertia_kernel, dim3((unsigned)ntile, (unsigned)n_init), dim3(256), 0, s, Xc, n, ds, C0, C1, st, labels, ipart);
  MMF_LAUNCH_CHECK();
  hipLaunchKernelGGL(km_pick_kernel, dim3(1), dim3(1024), 0, s, st, (int)n_init, n, k, ipart, ntile, labels, map, best);
  MMF_LAUNCH_CHECK();
  const int64_t fin = n > k * d ? n : k * d;
  hipLaunchKernelGGL(km_finish_kernel, dim3((unsigned)((fin + 255) / 256)), dim3(256), 0, s, st, best, n, k, d, ds, labels, C0, C1, mean32, out_labels,
                     out_centres);
  MMF_LAUNCH_CHECK();
  if (info_h) {
    // [0] best restart, [1] its inertia, [2] its iterations, [3] tol_abs, [4] ambiguous draws, [5] ambiguous trial choices,
    // [6] Lloyd lockstep iterations, then per restart: inertia, iterations
    std::vector<KmState> hs((size_t)n_init);
    int h_best = 0;
    unsigned int h_amb[2] = {0, 0};
    double h_tol = 0.0;
    MMF_HIP(hipMemcpyAsync(hs.data(), st, (size_t)n_init * sizeof(KmState), hipMemcpyDeviceToHost, s));
    MMF_HIP(hipMemcpyAsync(&h_best, best, 4, hipMemcpyDeviceToHost, s));
    MMF_HIP(hipMemcpyAsync(h_amb, amb, 8, hipMemcpyDeviceToHost, s));
    MMF_HIP(hipMemcpyAsync(&h_tol, tol_abs, 8, hipMemcpyDeviceToHost, s));
    MMF_HIP(hipStreamSynchronize(s));
    info_h[0] = (double)h_best; info_h[1] = hs[h_best].inertia; info_h[2] = (double)hs[h_best].n_iter; info_h[3] = h_tol;
    info_h[4] = (double)h_amb[0]; info_h[5] = (double)h_amb[1]; info_h[6] = (double)it;
    for (int64_t g = 0; g < n_init; ++g) { info_h[7 + 2 * g] = hs[g].inertia; info_h[8 + 2 * g] = (double)hs[g].n_iter; }
  }
  return MMF_OK;
}

}  // namespace mmf
