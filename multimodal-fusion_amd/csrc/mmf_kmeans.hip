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
#include <type_traits>

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
  int64_t b = 0;
  for (; b + 8 <= nblk; b += 8) {                     // eight partials in flight, added in order
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[(b + u) * d + col];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; b < nblk; ++b) s += partial[b * d + col];
  s /= (double)n;
  out64[col] = s;
  if (out32) out32[col] = (float)s;
}

// mean32[col] = (((x_0 + x_1) + x_2) + ... ) / float32(n): the float32 row-by-row accumulation numpy's X.mean(axis=0) performs
// on a C-contiguous float32 matrix with d >= 2 (for d = 1 the column is contiguous and numpy sums it pairwise: an ulp of
// difference in the mean there), so the centred data are scikit-learn's bit for bit.  A chain of n dependent adds per column is
// latency, not bandwidth: a workgroup owns 8 columns (d / 8 workgroups keep enough loads in flight), stages chunks of 512 rows
// through LDS — all 256 threads load, the next chunk's loads are in flight while the chain runs — and 8 lanes walk the rows.
constexpr int CM_COLS = 8, CM_ROWS = 512;
__global__ __launch_bounds__(256) void km_colmean_seq_kernel(const float* __restrict__ X, int64_t n, int64_t d, float* __restrict__ mean32) {
  __shared__ float tile[CM_ROWS][CM_COLS];
  const int c = threadIdx.x & (CM_COLS - 1), q = threadIdx.x / CM_COLS;      // q in [0, 32)
  const int64_t col = (int64_t)blockIdx.x * CM_COLS + c;
  const bool live = col < d;
  float regs[16];
  float acc = 0.f;
  const int64_t nchunks = (n + CM_ROWS - 1) / CM_ROWS;
  auto load = [&](int64_t ch) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t r = ch * CM_ROWS + q + 32 * i;
      regs[i] = (live && r < n) ? X[r * d + col] : 0.f;
    }
  };
  load(0);
  for (int64_t ch = 0; ch < nchunks; ++ch) {
#pragma unroll
    for (int i = 0; i < 16; ++i) tile[q + 32 * i][c] = regs[i];
    __syncthreads();
    if (ch + 1 < nchunks) load(ch + 1);
    if (threadIdx.x < CM_COLS) {
      const int64_t rows = (n - ch * CM_ROWS < CM_ROWS) ? (n - ch * CM_ROWS) : CM_ROWS;
      if (rows == CM_ROWS) {
        for (int r0 = 0; r0 < CM_ROWS; r0 += 32) {       // 32 LDS reads in flight, then the 32 dependent adds
          float v[32];
#pragma unroll
          for (int u = 0; u < 32; ++u) v[u] = tile[r0 + u][c];
#pragma unroll
          for (int u = 0; u < 32; ++u) acc += v[u];
        }
      } else {
        for (int r = 0; r < (int)rows; ++r) acc += tile[r][c];
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < CM_COLS && live) mean32[col] = acc / (float)n;
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

// rowp[2] / ptp[2]: this thread's two staging rows (row t >> 3 and 32 + (t >> 3) of the tile) at column 4 (t & 7).
// MT = 16-row MFMA tiles per wave.  Plain form: tile (16 MT) rows x 64 points, wave w owns points 16 w .. against all rows (MT < 4:
// the last, partly filled row tile of the E step — k = 100 is 64 + 48 rows, not 128).  SPLIT form (MT = 2): tile 64 rows x
// 32 points, wave w owns rows 32 (w & 1) .. and points 16 (w >> 1) .. (twice the workgroups when 64-point tiles would leave the
// chip with one wave per SIMD and nothing to hide the global loads behind).  Global loads run TWO chunks ahead of the MFMAs.
template <int MT, bool SPLIT>
__device__ __forceinline__ void km_tile_dots(const float* const rowp[2], const float* const ptp[2], int64_t nchunks, double* As, double* Bs,
                                             f64x4 acc[MT]) {
  static_assert(!SPLIT || MT == 2, "the split form is 2 x 16 rows per wave");
  constexpr int NB = SPLIT ? 1 : 2;          // staging rows of the point tile per thread (32 or 64 points)
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, g = lane >> 4;
  const int sr = t >> 3, sk = (t & 7) * 4;
  const int arow0 = SPLIT ? 32 * (w & 1) : 0;
  const int bpt = SPLIT ? 16 * (w >> 1) : 16 * w;
  float4 ra[2][2], rb[2][NB];
  auto gload = [&](int st, int64_t c) {
#pragma unroll
    for (int u = 0; u < 2; ++u) ra[st][u] = *reinterpret_cast<const float4*>(rowp[u] + c * KM_KC);
#pragma unroll
    for (int u = 0; u < NB; ++u) rb[st][u] = *reinterpret_cast<const float4*>(ptp[u] + c * KM_KC);
  };
  // piece `part` (0 .. 3) of writing register stage `st` into LDS buffer `buf`: the conversions and LDS writes of the next
  // chunk are spread between the MFMA groups of the current one, where they issue while the matrix pipe is busy
  auto swrite_part = [&](int st, int buf, int part) {
    double* An = As + buf * (KM_T * KM_LD);
    double* Bn = Bs + buf * (KM_T * KM_LD);
    if (part < 2) km_stage_store(An + (sr + 32 * part) * KM_LD + sk, ra[st][part]);
    else if (part - 2 < NB) km_stage_store(Bn + (sr + 32 * (part - 2)) * KM_LD + sk, rb[st][part - 2]);
  };
  // chunk in LDS buffer `buf` -> accumulators; behind MFMA group q, piece q of the staging of stage `st` into the other buffer
  auto compute = [&](int buf, int st) {
    const double* A = As + buf * (KM_T * KM_LD) + (arow0 + (lane & 15)) * KM_LD + 2 * g;
    const double* B = Bs + buf * (KM_T * KM_LD) + (bpt + (lane & 15)) * KM_LD + 2 * g;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f64x2 b = *reinterpret_cast<const f64x2*>(B + 8 * q);
      f64x2 a[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const f64x2*>(A + m * 16 * KM_LD + 8 * q);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].x, b.x, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].y, b.y, acc[m], 0, 0, 0);
      swrite_part(st, buf ^ 1, q);
    }
  };
  const int64_t lastc = nchunks - 1;
  gload(0, 0);
  gload(1, lastc < 1 ? lastc : 1);
#pragma unroll
  for (int part = 0; part < 4; ++part) swrite_part(0, 0, part);
  __syncthreads();
  // chunk c is in LDS buffer c & 1, chunk c + 1 in register stage (c + 1) & 1; chunk c + 2 is requested into stage c & 1.
  // No branches inside an iteration (chunk indices past the end are clamped: their loads and LDS writes are harmless), so the
  // staging interleaves with the MFMAs.
  for (int64_t c = 0; c < nchunks; c += 2) {
    gload(0, c + 2 < lastc ? c + 2 : lastc);      // stage 0 (chunk c) went into LDS during the previous iteration
    compute(0, 1);                               // reads buffer 0, stages chunk c + 1 (stage 1) into buffer 1
    __syncthreads();
    if (c + 1 >= nchunks) break;
    gload(1, c + 3 < lastc ? c + 3 : lastc);
    compute(1, 0);                               // reads buffer 1, stages chunk c + 2 (stage 0) into buffer 0
    __syncthreads();
  }
}

// Seeding distances: out[r][i] = min(clamp_r[i], float32(max(0, (-2 c_r.x_i + |c_r|^2) + |x_i|^2))) for the rows c_r = Xc[cand[r]],
// r < R, and partial[r][tile] = sum of out[r][i] over the tile's points (f64).  clamp_r = prev[sel[r / group]]: the running
// closest-centre distances of r's restart, which are a ROW OF THE PREVIOUS STEP'S OUTPUT (the trial that won): nothing is copied.
// prev == NULL: no clamp (the first centre).
template <int MT>
__global__ __launch_bounds__(256, 2) void km_seed_dots_kernel(const float* __restrict__ Xc, int64_t n, int64_t ds, const double* __restrict__ xx,
                                                              const int64_t* __restrict__ cand, int64_t R, int64_t group,
                                                              const float* __restrict__ prev, const int* __restrict__ sel,
                                                              float* __restrict__ out, double* __restrict__ partial) {
  constexpr int PT = (MT == 4) ? 64 : 32;
  extern __shared__ double km_lds[];
  double* As = km_lds;
  double* Bs = km_lds + 2 * KM_T * KM_LD;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int64_t p0 = (int64_t)blockIdx.x * PT, r0 = (int64_t)blockIdx.y * KM_T;
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
  // what the epilogue needs from memory is requested before the contraction: |c_r|^2 of this lane's rows, |x_p|^2 of its point
  constexpr bool SPLIT = MT == 2;
  const int arow0 = SPLIT ? 32 * (w & 1) : 0;
  const int bpt = SPLIT ? 16 * (w >> 1) : 16 * w;
  double cr[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      int64_t r = r0 + arow0 + 16 * m + 4 * v + (lane >> 4);
      if (r >= R) r = R - 1;
      cr[m][v] = xx[cand[r]];
    }
  const int64_t pq = p0 + bpt + (lane & 15);
  const double xp = xx[pq < n ? pq : n - 1];
  // ... and the clamp values of the 16 rows this wave finishes (wave w: rows 16 w .. 16 w + 15, lane = point)
  const int64_t p = p0 + lane;
  const bool live = lane < PT && p < n;
  float cl[16];
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    const int64_t r = r0 + 16 * w + rr;
    cl[rr] = __builtin_huge_valf();
    if (prev && live && r < R) cl[rr] = prev[(int64_t)sel[r / group] * n + p];
  }
  f64x4 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = f64x4{0.0, 0.0, 0.0, 0.0};
  km_tile_dots<MT, MT == 2>(rowp, ptp, ds / KM_KC, As, Bs, acc);
  // epilogue through LDS: [64 rows][PT + 1] f32
  float* T = reinterpret_cast<float*>(km_lds);
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int rl = arow0 + 16 * m + 4 * v + (lane >> 4);
      double dd = -2.0 * acc[m][v];
      dd = dd + cr[m][v];
      dd = dd + xp;
      float f = (float)dd;
      f = f > 0.f ? f : 0.f;
      T[rl * (PT + 1) + bpt + (lane & 15)] = f;
    }
  __syncthreads();
  // wave w finishes rows 16 w .. 16 w + 15: clamp, store, tile sum
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    const int rl = 16 * w + rr;
    const int64_t r = r0 + rl;
    double sm = 0.0;
    if (live && r < R) {
      float f = T[rl * (PT + 1) + lane];
      f = f < cl[rr] ? f : cl[rr];
      out[r * n + p] = f;
      sm = (double)f;
    }
    sm = wave_sum(sm);
    if (lane == 0 && r < R) partial[r * gridDim.x + blockIdx.x] = sm;
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
    // the row tile's height in 16-row MFMA tiles: only the last tile of a restart's centres is partly filled
    const int64_t left = k - r0;
    auto tile = [&](auto mt_tag) {
      constexpr int MT = decltype(mt_tag)::value;
      f64x4 acc[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = f64x4{0.0, 0.0, 0.0, 0.0};
      km_tile_dots<MT, false>(rowp, ptp, ds / KM_KC, As, Bs, acc);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int64_t r = r0 + 16 * m + 4 * v + (lane >> 4);     // ascending in (m, v) for a lane: strict < keeps the first minimum
          if (r < k) {
            const double s = ccg[r] - 2.0 * acc[m][v];
            if (s < best) { best = s; bi = (int)r; }
          }
        }
    };
    if (left > 48) tile(std::integral_constant<int, 4>{});
    else if (left > 32) tile(std::integral_constant<int, 3>{});
    else if (left > 16) tile(std::integral_constant<int, 2>{});
    else tile(std::integral_constant<int, 1>{});
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

// One workgroup per restart, between two launches of the distance kernel.
// (1) CHOICE among the `tprev` trials of the step that has just been computed: potential of a trial = float32 of the f64 sum of
//     its tile partials; the first smallest wins (np.argmin over float32 potentials).  Its candidate becomes centre `step_prev` of
//     the restart, its row of `rows` (index sel[g]) the restart's running closest-centre distances.
// (2) DRAW of the next step's `trials` candidates (U != NULL): candidate = searchsorted(cumsum(closest as f64), u * float64(pot32),
//     side='left'), clipped to n - 1.  The cumulative sum is never formed: the tile partials of the chosen row ARE its block sums —
//     a thread owns a run of consecutive tiles, the 256 run totals are scanned in LDS, a target finds its run by binary search, its
//     tile by walking the run's partials and its point by walking the tile's `pt` values.
// amb[0] counts draws that land within 4 float32 ulps of the potential of a boundary of the cumulative sum, amb[1] choices whose
// runner-up (a different point) is within 4 float32 ulps: decisions scikit-learn's own float32 BLAS sums may take either way.
constexpr int KM_STEP_CACHE = 6144;      // tile partials (f64) of all trials of a restart kept in LDS when they fit (48 KiB)
__global__ __launch_bounds__(256) void km_seed_step_kernel(const float* __restrict__ rows, const double* __restrict__ partial, int64_t ntile, int pt,
                                                           int64_t n, int tprev, const int64_t* __restrict__ cand_prev, int64_t k, int64_t step_prev,
                                                           const double* __restrict__ U, int64_t u_stride, int trials, float* __restrict__ pot32,
                                                           int64_t* __restrict__ seeds, int* __restrict__ sel, int64_t* __restrict__ cand_next,
                                                           unsigned int* __restrict__ amb) {
  // What this kernel reads was written by the launch before it, mostly on other XCDs: every DEPENDENT global access is a
  // round trip of 1 - 2 us.  So: one round for everything whose address is known at entry (all trials' tile partials into LDS,
  // the uniforms, the candidates), the choice and the search out of LDS, one more round for the chosen tile's values.
  __shared__ float pots[64];
  __shared__ double runs[16][64];               // inclusive scan of the run totals, per trial (tprev <= 64: 16 at a time)
  __shared__ double tp[KM_STEP_CACHE];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int64_t g = blockIdx.x;
  const bool cached = (int64_t)tprev * ntile <= KM_STEP_CACHE && tprev <= 16;
  const int64_t per = (ntile + 63) / 64, b = (int64_t)lane * per;
  int64_t e = b + per;
  if (e > ntile) e = ntile;
  double u_mine = 0.0;
  int64_t cand_mine = 0;
  if (U && w == 0 && lane < trials) u_mine = U[g * u_stride + lane];
  if (w == 0 && lane < tprev) cand_mine = cand_prev[g * tprev + lane];
  for (int tr = w; tr < tprev; tr += 4) {       // wave w: trials w, w + 4, ...: lane l owns the run of tiles [l per, (l + 1) per)
    const double* p = partial + (g * tprev + tr) * ntile;
    double sum = 0.0;
    for (int64_t j = b; j < e; ++j) {
      const double v = p[j];
      if (cached) tp[(int64_t)tr * ntile + j] = v;
      sum += v;
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {           // inclusive scan over the lanes
      const double v = __shfl_up(sum, o);
      if (lane >= o) sum += v;
    }
    if (tr < 16) runs[tr][lane] = sum;
    if (lane == 63) pots[tr] = (float)sum;
  }
  __syncthreads();
  if (w != 0) return;
  int bsel = 0;
  for (int tr = 1; tr < tprev; ++tr)
    if (pots[tr] < pots[bsel]) bsel = tr;
  const int64_t cand_best = __shfl(cand_mine, bsel);
  {
    const bool close = lane < tprev && lane != bsel && cand_mine != cand_best && pots[lane] - pots[bsel] <= 4.f * 1.1920928955078125e-07f * pots[bsel];
    const unsigned long long any = __ballot(close);
    if (lane == 0) {
      if (any) atomicAdd(amb + 1, 1u);
      seeds[g * k + step_prev] = cand_best;
      pot32[g] = pots[bsel];
      sel[g] = (int)(g * tprev + bsel);
    }
  }
  if (!U) return;
  // the draw: candidate = searchsorted(cumsum(chosen row), u * float64(pot32), 'left') through run totals -> tile partials -> values
  const int64_t row = g * tprev + bsel;
  const double* bp = partial + row * ntile;
  const float* cl = rows + row * n;
  if (bsel >= 16) {                              // (never with scikit-learn's 2 + log k trials) scan the chosen row again
    double sum = 0.0;
    for (int64_t j = b; j < e; ++j) sum += bp[j];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double v = __shfl_up(sum, o);
      if (lane >= o) sum += v;
    }
    runs[0][lane] = sum;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
  const double* rn = runs[bsel < 16 ? bsel : 0];
  if (lane < trials) {
    const double pot = (double)pots[bsel];
    const double target = u_mine * pot;
    int lo = 0, hi = 63;                        // first run whose inclusive total reaches the target
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (rn[mid] >= target) hi = mid; else lo = mid + 1;
    }
    const bool beyond = !(rn[63] >= target);    // the target lies above the whole sum (float32 rounding of the potential): n - 1
    double run = lo ? rn[lo - 1] : 0.0;
    int64_t tile = (int64_t)lo * per, tlast = tile + per;
    if (tlast > ntile) tlast = ntile;
    for (; tile < tlast - 1; ++tile) {           // the tile inside the run
      const double nx = run + (cached ? tp[(int64_t)bsel * ntile + tile] : bp[tile]);
      if (nx >= target) break;
      run = nx;
    }
    if (tile > ntile - 1) tile = ntile - 1;
    int64_t pick = n - 1;
    double below = run, at = run;
    if (!beyond) {
      // the tile's values first (all loads in flight), then the walk
      const int64_t j0 = tile * pt;
      float v[64];
#pragma unroll
      for (int u = 0; u < 64; ++u) v[u] = (u < pt && j0 + u < n) ? cl[j0 + u] : 0.f;
      bool found = false;
#pragma unroll
      for (int u = 0; u < 64; ++u) {
        if (!found && u < pt && j0 + u < n) {
          below = run;
          run += (double)v[u];
          at = run;
          if (run >= target) { pick = j0 + u; found = true; }
        }
      }
      if (!found) {                              // rounding between the partial and the element sums: continue into the following tiles
        for (int64_t j = j0 + pt; j < n; ++j) {
          below = run;
          run += (double)cl[j];
          at = run;
          if (run >= target) { pick = j; break; }
        }
      }
    }
    const double band = 4.0 * 1.1920928955078125e-07 * pot;
    if (target - below <= band || at - target <= band) atomicAdd(amb, 1u);
    cand_next[g * trials + lane] = pick;
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
    // eight row loads in flight per wave; the member indices of the NEXT trip are read while this trip's rows arrive (index -> row
    // is two dependent round trips otherwise).  The adds stay in member order.
    int64_t q = b + w;
    int64_t r[8];
    bool more = q + 28 < e;
    if (more) {
#pragma unroll
      for (int u = 0; u < 8; ++u) r[u] = order[q + 4 * u] - base;
    }
    while (more) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = Xc[r[u] * ds + col];
      q += 32;
      more = q + 28 < e;
      if (more) {
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = order[q + 4 * u] - base;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += (double)x[u];
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

// One wave per restart, after the E (+ M) step of Lloyd iteration `it`: scikit-learn's convergence logic
// (_kmeans_single_lloyd): swap the centre buffers; unchanged labels -> done, no further E step; shift <= tol or the last
// iteration -> one more E step with the new centres.  A restart with empty clusters waits for the host (KM_RELOC).
__global__ __launch_bounds__(64) void km_state_kernel(KmState* __restrict__ st, int n_init, int64_t k, int64_t strips, const double* __restrict__ shift_part,
                                                      const double* __restrict__ tol_abs, int it, int max_iter, int only, int* __restrict__ status) {
  const int g = blockIdx.x, lane = threadIdx.x;
  if (only < 0 || only == g) {
    KmState s = st[g];
    if (s.state == KM_FINAL) {
      s.state = KM_DONE;
    } else if (s.state == KM_RUN || s.state == KM_RELOC) {
      if (s.state == KM_RUN && s.empty > 0) {
        s.state = KM_RELOC;
      } else {
        double sh = 0.0;
        for (int64_t q = lane; q < k * strips; q += 64) sh += shift_part[(int64_t)g * k * strips + q];
        sh = wave_sum(sh);
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
    if (lane == 0) st[g] = s;
    if (lane == 0) status[g] = s.state;
  } else if (lane == 0) {
    status[g] = st[g].state;
  }
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
  b += 2 * al((size_t)R * n * 4);              // trial rows, two steps
  b += 2 * al((size_t)R * 2 * ntile * 8);      // their tile partials (32-point tiles at most)
  b += 2 * al((size_t)R * 8) + al((size_t)n_init * 8) + al((size_t)n_init * 4) + al((size_t)n_init * 4);   // cand x 2, first, pot32, sel
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
  float* rows[2];
  double* rpart[2];
  int64_t* cand[2];
  for (int u = 0; u < 2; ++u) rows[u] = (float*)take((size_t)R * n * 4);
  for (int u = 0; u < 2; ++u) rpart[u] = (double*)take((size_t)R * 2 * ntile * 8);
  for (int u = 0; u < 2; ++u) cand[u] = (int64_t*)take((size_t)R * 8);
  int64_t* first = (int64_t*)take((size_t)n_init * 8);
  float* pot32 = (float*)take((size_t)n_init * 4);
  int* sel = (int*)take((size_t)n_init * 4);
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
  MMF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(km_seed_dots_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  MMF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(km_seed_dots_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
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
  hipLaunchKernelGGL(km_colmean_seq_kernel, dim3((unsigned)((d + CM_COLS - 1) / CM_COLS)), dim3(256), 0, s, X, n, d, mean32);
  MMF_LAUNCH_CHECK();
  hipLaunchKernelGGL(km_colsum_kernel, cgrid, dim3(256), 0, s, X, n, d, (const double*)mean64, colpart);
  MMF_LAUNCH_CHECK();
  hipLaunchKernelGGL(km_colfin_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, s, colpart, nblk, n, d, var64, (float*)nullptr);
  MMF_LAUNCH_CHECK();
  hipLaunchKernelGGL(km_tol_kernel, dim3(1), dim3(1024), 0, s, var64, d, tol, tol_abs);
  MMF_LAUNCH_CHECK();
  hipLaunchKernelGGL(km_centre_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, X, n, d, ds, mean32, Xc, xx);
  MMF_LAUNCH_CHECK();

  // k-means++ for all restarts in lockstep: per step one small launch (choice of the previous step's trials + draw of the next
  // candidates) and one launch of the distance kernel; the running closest-centre distances are a row of the previous step's output
  hipLaunchKernelGGL(km_seed_first_kernel, dim3((unsigned)((n_init + 63) / 64)), dim3(64), 0, s, first, n_init, k, seeds, cand[0]);
  MMF_LAUNCH_CHECK();
  const bool small_tiles = ntile < 512;          // 64-point tiles would not give every SIMD two waves: 32-point tiles
  const int pt = small_tiles ? 32 : 64;
  const int64_t stile = (n + pt - 1) / pt;
  auto dots = [&](const int64_t* cd, int64_t Rn, int64_t group, const float* prev, float* out, double* part) {
    const dim3 grid((unsigned)stile, (unsigned)((Rn + KM_T - 1) / KM_T));
    if (small_tiles) hipLaunchKernelGGL(km_seed_dots_kernel<2>, grid, dim3(256), lds, s, Xc, n, ds, xx, cd, Rn, group, prev, sel, out, part);
    else hipLaunchKernelGGL(km_seed_dots_kernel<4>, grid, dim3(256), lds, s, Xc, n, ds, xx, cd, Rn, group, prev, sel, out, part);
  };
  dots(cand[0], n_init, 1, nullptr, rows[0], rpart[0]);
  MMF_LAUNCH_CHECK();
  int cur = 0, tprev = 1;
  for (int64_t step = 1; step <= k; ++step) {
    const bool last = step == k;
    hipLaunchKernelGGL(km_seed_step_kernel, dim3((unsigned)n_init), dim3(256), 0, s, rows[cur], rpart[cur], stile, pt, n, tprev, cand[cur], k, step - 1,
                       last ? (const double*)nullptr : U + (step - 1) * trials, (k - 1) * trials, trials, pot32, seeds, sel, cand[cur ^ 1], amb);
    MMF_LAUNCH_CHECK();
    if (last) break;
    dots(cand[cur ^ 1], R, trials, rows[cur], rows[cur ^ 1], rpart[cur ^ 1]);
    MMF_LAUNCH_CHECK();
    cur ^= 1;
    tprev = trials;
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
    hipLaunchKernelGGL(km_state_kernel, dim3((unsigned)n_init), dim3(64), 0, s, st, (int)n_init, k, strips, shift_part, tol_abs, it, max_iter,
                       -1, status);
    MMF_LAUNCH_CHECK();
    MMF_HIP(hipMemcpyAsync(h_status.data(), status, (size_t)n_init * 4, hipMemcpyDeviceToHost, s));
    MMF_HIP(hipStreamSynchronize(s));
    for (int64_t g = 0; g < n_init; ++g) {
      if (h_status[g] != KM_RELOC) continue;
      hipLaunchKernelGGL(km_relocate_kernel, dim3(1), dim3(1024), 0, s, Xc, n, ds, k, (int)g, labels, offsets, C0, C1, sums32, st, dist, cntf, shift_part,
                         strips);
      MMF_LAUNCH_CHECK();
      hipLaunchKernelGGL(km_state_kernel, dim3((unsigned)n_init), dim3(64), 0, s, st, (int)n_init, k, strips, shift_part, tol_abs, it, max_iter,
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
