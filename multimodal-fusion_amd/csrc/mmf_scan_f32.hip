// mmf_scan_f32.hip — exact all-pairs scan on v_mfma_f32_32x32x2_f32, any d, f32/bf16/f16 inputs.
//
// One kernel template, two epilogues:
//   MODE_SCAN  : fused similarity + per-row top-k candidates (the N x M matrix never leaves the CU)
//   MODE_DENSE : writes the [n,m] similarity for the small-N reference signatures
//
// The f32-input MFMA accumulates D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)), i.e. exactly the k-ordered
// fmaf chain of include/mmf_hg.h (cdna_hip_programming.md §3 "FP32-input MFMA"), so the keys formed
// here ARE the canonical keys: lists are truncated by the final total order and cannot overflow.
//
// Tiling: workgroup = 4 waves = 128 queries x 128 candidates per macro tile, K staged through LDS in
// chunks of 32 (double buffered, one barrier per chunk).  Wave w owns queries 32w..32w+31 as the MFMA
// column (B operand) and sweeps the 4 candidate sub-tiles as MFMA rows (A operand), so every lane
// keeps ONE query and its candidate list is lane-private (mmf_dev.h).
// LDS rows are padded to KC + 1 floats: lanes 0..31 of a ds_read_b32 hit 32 distinct banks.  (An 8-byte image —
// k0 k2 | k1 k3 inside every group of four, one ds_read_b64 per operand per two k-steps, ds_write_b64 staging —
// halves the LDS instructions and measured 6 % SLOWER in a same-process A/B, round 2: the 4-byte image stays.)
//
// Replaces: torch.mm + 3 elementwise passes, build_hypergraph/similarity_kernel.py:43-52, 79-84, 122;
//           sklearn brute-force kneighbors, build_hypergraph/preprocess_hypergraph.py:379-382.
#include <stdlib.h>

#include "mmf_dev.h"
#include "mmf_host.h"

namespace mmf {

constexpr int F_NT = 256;
constexpr int F_QT = 128;
constexpr int F_CT = 128;
// k per staged chunk (template parameter KC, row stride KC + 1 floats): 16 for the scan — 2 x (128 + 128) rows x 17
// floats = 35 KB beside 32 KB of lists, so two workgroups share a CU — and 32 for the dense outputs (no lists: two
// workgroups fit anyway, and half the barriers measure 7 % faster).

constexpr int MODE_SCAN = 0;
constexpr int MODE_DENSE = 1;

struct ScanF32Args {
  const void* X; const void* Y;
  int64_t n, m, d;
  int dtype;
  const float* rx; const float* cy;
  const int32_t* row_ids; int64_t n_rows;
  float neg_lambda;
  int metric;
  int kk;
  int col_splits;
  int64_t tiles_per_split;
  uint32_t* cand_cnt; uint32_t* cand_ids; uint32_t* overflow;
  // dense epilogue
  float* out;
  const float* P; int dp; float neg_lambda_g;
  int64_t prow0;   // dense combined, panel form: query q of this launch is row prow0 + q of P
  int debug;   // MMF_F32_DEBUG (timing-only ablations): 1 skip epilogue, 2 skip staging
};

template <bool VEC4>
__device__ __forceinline__ f32x4 load4(const void* base, int64_t row, int64_t k, int64_t d, int dtype) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if constexpr (VEC4) {
    if (k < d) v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + row * d + k);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (k + i < d) v[i] = ld_elem(base, row * d + k + i, dtype);
  }
  return v;
}

template <int MODE, int CAP, bool VEC4, int F_KC>
__global__ __launch_bounds__(F_NT, (MODE == MODE_DENSE || CAP <= 16) ? 2 : 1) void scan_f32_kernel(ScanF32Args a) {
  constexpr int F_LD = F_KC + 1;
  constexpr int TPR = F_KC / 4;          // staging: threads per row (16 bytes each)
  constexpr int RPP = F_NT / TPR;        //          rows per pass
  constexpr int NP = F_QT / RPP;         //          passes per 128-row tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Qs = reinterpret_cast<float*>(smem);   // [2][F_QT][F_LD]
  float* Cs = Qs + 2 * F_QT * F_LD;             // [2][F_CT][F_LD]
  float* cys = Cs + 2 * F_CT * F_LD;            // [2][F_CT] per-candidate scalars of the tile being accumulated
  float* lkeys = cys + 2 * F_CT;                // [CAP][F_NT]
  uint32_t* lids = reinterpret_cast<uint32_t*>(lkeys + CAP * F_NT);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int c = lane & 31;

  const int split = blockIdx.x % a.col_splits;
  const int64_t rb = blockIdx.x / a.col_splits;
  const int64_t q0 = rb * F_QT;

  const int64_t total_tiles = (a.m + F_CT - 1) / F_CT;
  int64_t t_begin = (int64_t)split * a.tiles_per_split;
  int64_t t_end = t_begin + a.tiles_per_split;
  if (t_end > total_tiles) t_end = total_tiles;
  if (t_begin > t_end) t_begin = t_end;
  const int nkc = (int)((a.d + F_KC - 1) / F_KC);
  const int64_t steps = (t_end - t_begin) * nkc;

  // this lane's query
  const int64_t qpos = q0 + 32 * wave + c;
  const bool qvalid = qpos < a.n_rows;
  const int64_t qrow = qvalid ? (a.row_ids ? (int64_t)a.row_ids[qpos] : qpos) : 0;
  const float ri = qvalid ? a.rx[qrow] : 1.0f;
  const int metric = a.metric;

  // DENSE: the 16 query rows this lane's accumulator elements belong to (fixed for the workgroup's lifetime)
  float rq16[16];
  if constexpr (MODE == MODE_DENSE) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t q = q0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * half;
      rq16[r] = (q < a.n_rows) ? a.rx[q] : 1.0f;
    }
  }

  LaneList<CAP, F_NT> list;
  if constexpr (MODE == MODE_SCAN) {
    list.init(lkeys + tid, lids + tid);
    if (!qvalid) list.thr = __builtin_huge_valf();
  }

  // staging roles: rows (tid / TPR) + RPP*i, k offset 4*(tid % TPR)
  const int srow = tid / TPR;
  const int sk = 4 * (tid % TPR);
  int64_t qsrc[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    int64_t p = q0 + srow + RPP * i;
    if (p > a.n_rows - 1) p = a.n_rows - 1;
    qsrc[i] = a.row_ids ? (int64_t)a.row_ids[p] : p;
  }

  f32x4 rq[NP], rc[NP];
  float rcy = 0.0f;
  auto gload = [&](int64_t step) {
    const int64_t ct = t_begin + step / nkc;
    const int64_t k = (int64_t)(step % nkc) * F_KC + sk;
    if ((step % nkc) == 0 && tid < F_CT) {       // first chunk of a tile: its 128 per-candidate scalars ride along
      int64_t j = ct * F_CT + tid;
      if (j > a.m - 1) j = a.m - 1;
      rcy = a.cy[j];
    }
    const int64_t kbase = (int64_t)(step % nkc) * F_KC;
    if (VEC4 && kbase + F_KC <= a.d) {          // full chunk (wave-uniform): plain 16-byte loads, no predicates
      const float* xf = reinterpret_cast<const float*>(a.X);
      const float* yf = reinterpret_cast<const float*>(a.Y);
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        rq[i] = *reinterpret_cast<const f32x4*>(xf + qsrc[i] * a.d + k);
        int64_t j = ct * F_CT + srow + RPP * i;
        if (j > a.m - 1) j = a.m - 1;
        rc[i] = *reinterpret_cast<const f32x4*>(yf + j * a.d + k);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        rq[i] = load4<VEC4>(a.X, qsrc[i], k, a.d, a.dtype);
        int64_t j = ct * F_CT + srow + RPP * i;
        if (j > a.m - 1) j = a.m - 1;
        rc[i] = load4<VEC4>(a.Y, j, k, a.d, a.dtype);
      }
    }
  };
  auto swrite = [&](int buf, int64_t step) {
    if ((step % nkc) == 0 && tid < F_CT) cys[((step / nkc) & 1) * F_CT + tid] = rcy;
    float* qd = Qs + buf * F_QT * F_LD + srow * F_LD + sk;
    float* cd = Cs + buf * F_CT * F_LD + srow * F_LD + sk;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        qd[i * RPP * F_LD + e] = rq[i][e];
        cd[i * RPP * F_LD + e] = rc[i][e];
      }
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  if (steps > 0) {
    gload(0);
    swrite(0, 0);
  }
  __syncthreads();

  for (int64_t s = 0; s < steps; ++s) {
    const int buf = (int)(s & 1);
    if (s + 1 < steps && !(a.debug & 2)) gload(s + 1);

    const float* Qb = Qs + buf * F_QT * F_LD + (32 * wave + c) * F_LD + half;
    const float* Cb = Cs + buf * F_CT * F_LD + c * F_LD + half;
    // operands of k-step k2+1 are read from LDS before the MFMAs of k-step k2 issue (one wave per SIMD:
    // nothing else hides the LDS latency); the interleave is pinned for the scheduler
    float bn = Qb[0];
    float avn[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) avn[t] = Cb[t * 32 * F_LD];
#pragma unroll
    for (int k2 = 0; k2 < F_KC / 2; ++k2) {
      const float b = bn;
      float av[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) av[t] = avn[t];
      if (k2 + 1 < F_KC / 2) {
        bn = Qb[2 * (k2 + 1)];
#pragma unroll
        for (int t = 0; t < 4; ++t) avn[t] = Cb[t * 32 * F_LD + 2 * (k2 + 1)];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        // SCAN: C rows = candidates, columns = queries (a lane owns one query).  DENSE: the transpose — a lane owns
        // one candidate COLUMN of the output, so the 32 lanes of a half store 128 contiguous bytes of one output
        // row.  Products commute, the k order is the same: both orientations give the same bits.
        if constexpr (MODE == MODE_DENSE) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, av[t], acc[t], 0, 0, 0);
        else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], b, acc[t], 0, 0, 0);
      }
      if (k2 + 1 < F_KC / 2) __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }

    if ((int)(s % nkc) == nkc - 1 && !(a.debug & 1)) {
      const int64_t ct = t_begin + s / nkc;
      // one 32 x 32 sub-tile at a time (keeping all four key vectors live costs 64 VGPRs and pushes the
      // staging registers of the next chunk into AGPRs, i.e. a vmcnt(0) in front of the MFMA block)
      if constexpr (MODE == MODE_DENSE) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int64_t j = ct * F_CT + 32 * t + c;            // this lane's output column
          const bool jv = j < a.m;
          const float cj = cys[((s / nkc) & 1) * F_CT + 32 * t + c];
          const float nl = (metric == MMF_RBF) ? a.neg_lambda : -1.0f;  // (-1)*sq == -sq exactly
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t q = q0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float dot = acc[t][r];
            acc[t][r] = 0.0f;
            if (jv && q < a.n_rows) {
              float key;
              if (metric == MMF_DOT) key = dot;
              else if (metric == MMF_COSINE) key = key_from_dot<MMF_COSINE>(dot, rq16[r], cj, 0.0f);
              else key = key_from_dot<MMF_RBF>(dot, rq16[r], cj, nl);
              float v = (metric == MMF_RBF) ? expf(key) : key;
              if (a.P) {
                // K_g from the positions, canonical chains over dp (similarity_kernel.py:79-84, 122)
                float ni = 0.f, nj = 0.f, dp_ = 0.f;
                for (int e = 0; e < a.dp; ++e) {
                  const float pi = a.P[(a.prow0 + q) * a.dp + e], pj = a.P[j * a.dp + e];
                  ni = __builtin_fmaf(pi, pi, ni);
                  nj = __builtin_fmaf(pj, pj, nj);
                  dp_ = __builtin_fmaf(pi, pj, dp_);
                }
                v = v * expf(a.neg_lambda_g * sq_from(ni, nj, dp_));
              }
              a.out[q * a.m + j] = v;
            }
          }
        }
      } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int64_t cand0 = ct * F_CT + 32 * t;
        const float* cyt = cys + ((s / nkc) & 1) * F_CT + 32 * t + 4 * half;
        f32x16 key;
        // keys in three wave-uniform flavours (no per-element switch): dot | dot/(r*c) | nl*((r+c)-2dot)
        if (metric == MMF_DOT) {
#pragma unroll
          for (int r = 0; r < 16; ++r) key[r] = acc[t][r];
        } else if (metric == MMF_COSINE) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 c4 = *reinterpret_cast<const f32x4*>(cyt + 8 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) key[4 * g + i] = key_from_dot<MMF_COSINE>(acc[t][4 * g + i], ri, c4[i], 0.0f);
          }
        } else {
          const float nl = (metric == MMF_RBF) ? a.neg_lambda : -1.0f;  // (-1)*sq == -sq exactly
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 c4 = *reinterpret_cast<const f32x4*>(cyt + 8 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) key[4 * g + i] = key_from_dot<MMF_RBF>(acc[t][4 * g + i], ri, c4[i], nl);
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool jv = (cand0 + (r & 3) + 8 * (r >> 2) + 4 * half) < a.m;
          key[r] = jv ? key[r] : kNegInf;
          acc[t][r] = 0.0f;
        }
        if (__builtin_expect(__any(max16(key) >= list.thr), 0)) list.template offer_tile<true>(key, (uint32_t)cand0, half, a.kk, 0.0f);
      }
    }

      }
    if (s + 1 < steps && !(a.debug & 2)) swrite(buf ^ 1, s + 1);
    __syncthreads();
  }

  if constexpr (MODE == MODE_SCAN) {
    list.template compact<true>(a.kk, 0.0f);
    if (qvalid) {
      const int64_t lbase = qpos * (2 * a.col_splits) + 2 * split + half;
      a.cand_cnt[lbase] = (uint32_t)list.cnt;
      for (int e = 0; e < list.cnt; ++e) a.cand_ids[lbase * CAP + e] = list.ids[e * F_NT];
      if (list.overflow) atomicOr(a.overflow + qpos, 1u);
    }
  }
}

// ------------------------------------------------------------------------------------------------
int scan_f32_cap(int kk) {
  if (kk <= 12) return 16;
  if (kk <= 28) return 32;
  if (kk <= 44) return 48;     // 48 x 256 x 8 B of lists + 36 KB of tiles: one workgroup per CU
  return 0;
}

static size_t scan_f32_lds(int cap, int kc) {
  return sizeof(float) * (2 * F_QT * (kc + 1) + 2 * F_CT * (kc + 1) + 2 * F_CT) + (size_t)cap * F_NT * 8;
}

template <int MODE, int CAP>
static int launch_f32_t(const ScanF32Args& a, bool vec4, int64_t grid, hipStream_t s) {
  if (a.metric < MMF_DOT || a.metric > MMF_RBF) {
    set_error("scan_f32: unsupported metric %d", a.metric);
    return MMF_E_INVALID;
  }
  constexpr int KC = (MODE == MODE_SCAN) ? 16 : 32;
  const size_t lds = scan_f32_lds(MODE == MODE_SCAN ? CAP : 0, KC);
  auto go = [&](auto kern) -> int {
    MMF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(F_NT), lds, s, a);
    MMF_LAUNCH_CHECK();
    return MMF_OK;
  };
  if (vec4) return go(scan_f32_kernel<MODE, CAP, true, KC>);
  return go(scan_f32_kernel<MODE, CAP, false, KC>);
}

static bool can_vec4(const void* X, const void* Y, int64_t d, int dtype) {
  return dtype == MMF_F32 && (d % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0) &&
         ((reinterpret_cast<uintptr_t>(Y) & 15) == 0);
}

int launch_scan_f32(const ScanProblem& p, const CandLists& L, hipStream_t s, int* grid_out) {
  if (p.n_rows <= 0 || p.m <= 0) return MMF_OK;
  ScanF32Args a{};
  a.X = p.X; a.Y = p.Y; a.n = p.n; a.m = p.m; a.d = p.d; a.dtype = p.dtype;
  a.rx = p.rx; a.cy = p.cy; a.row_ids = p.row_ids; a.n_rows = p.n_rows;
  a.neg_lambda = -p.lambda; a.kk = p.kk; a.col_splits = p.col_splits;
  const int64_t total_tiles = (p.m + F_CT - 1) / F_CT;
  a.tiles_per_split = (total_tiles + p.col_splits - 1) / p.col_splits;
  a.cand_cnt = L.cnt; a.cand_ids = L.ids; a.overflow = L.overflow;
  const int64_t grid = ((p.n_rows + F_QT - 1) / F_QT) * p.col_splits;
  if (grid_out) *grid_out = (int)grid;
  const bool v4 = can_vec4(p.X, p.Y, p.d, p.dtype);
  a.metric = p.metric;
  { const char* e = getenv("MMF_F32_DEBUG"); a.debug = e ? atoi(e) : 0; }
  if (L.cap == 16) return launch_f32_t<MODE_SCAN, 16>(a, v4, grid, s);
  if (L.cap == 32) return launch_f32_t<MODE_SCAN, 32>(a, v4, grid, s);
  if (L.cap == 48) return launch_f32_t<MODE_SCAN, 48>(a, v4, grid, s);
  set_error("scan_f32: unsupported list capacity %d", L.cap);
  return MMF_E_INTERNAL;
}

// d <= 8 (the spatial similarity of similarity_kernel.py:58-86 has d = 2): no matrix cores, the kernel is pure
// output bandwidth.  A thread owns 4 consecutive output columns (its 4 y vectors stay in registers) and walks 16
// rows; each store is 16 bytes per lane, 1 KiB contiguous per wave.  Same canonical chains and keys as everywhere.
__global__ __launch_bounds__(256) void dense_small_d_kernel(const void* __restrict__ X, int64_t n, const void* __restrict__ Y,
                                                            int64_t m, int d, int dtype, int metric, float neg_lambda,
                                                            const float* __restrict__ rx, const float* __restrict__ cy,
                                                            float* __restrict__ out) {
  const int64_t j0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const int64_t i0 = (int64_t)blockIdx.y * 16;
  if (j0 >= m) return;
  float y[4][8], cj[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int64_t j = (j0 + jj < m) ? (j0 + jj) : (m - 1);
    cj[jj] = cy[j];
#pragma unroll
    for (int k = 0; k < 8; ++k) y[jj][k] = (k < d) ? ld_elem(Y, j * d + k, dtype) : 0.0f;
  }
  const float nl = (metric == MMF_RBF) ? neg_lambda : -1.0f;
  const bool vec = ((m & 3) == 0) && (j0 + 3 < m) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  for (int ii = 0; ii < 16; ++ii) {
    const int64_t i = i0 + ii;
    if (i >= n) break;
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = (k < d) ? ld_elem(X, i * d + k, dtype) : 0.0f;
    const float ri = rx[i];
    f32x4 v;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      float dot = 0.0f;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < d) dot = __builtin_fmaf(x[k], y[jj][k], dot);
      float key;
      if (metric == MMF_DOT) key = dot;
      else if (metric == MMF_COSINE) key = key_from_dot<MMF_COSINE>(dot, ri, cj[jj], 0.0f);
      else key = key_from_dot<MMF_RBF>(dot, ri, cj[jj], nl);
      v[jj] = (metric == MMF_RBF) ? expf(key) : key;
    }
    float* o = out + i * m + j0;
    if (vec) *reinterpret_cast<f32x4*>(o) = v;
    else {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        if (j0 + jj < m) o[jj] = v[jj];
    }
  }
}

int launch_sim_dense(const void* X, int64_t n, const void* Y, int64_t m, int64_t d, int dtype, int metric,
                     float lambda, const float* rx, const float* cy, float* out, hipStream_t s) {
  if (n <= 0 || m <= 0) return MMF_OK;
  if (d <= 8 && metric != MMF_RBF_DIRECT) {
    dim3 grid((unsigned)((m + 1023) / 1024), (unsigned)((n + 15) / 16));
    hipLaunchKernelGGL(dense_small_d_kernel, grid, dim3(256), 0, s, X, n, Y, m, (int)d, dtype, metric, -lambda, rx, cy, out);
    MMF_LAUNCH_CHECK();
    return MMF_OK;
  }
  if (metric == MMF_RBF_DIRECT) return launch_rbf_direct(X, n, Y, m, d, dtype, lambda, out, nullptr, nullptr, s);   // mmf_direct.hip
  ScanF32Args a{};
  a.X = X; a.Y = Y; a.n = n; a.m = m; a.d = d; a.dtype = dtype;
  a.rx = rx; a.cy = cy; a.row_ids = nullptr; a.n_rows = n;
  a.neg_lambda = -lambda; a.kk = 0;
  const int64_t total_tiles = (m + F_CT - 1) / F_CT;
  const int64_t rbs = (n + F_QT - 1) / F_QT;
  int64_t splits = (1024 + rbs - 1) / rbs;
  if (splits > total_tiles) splits = total_tiles;
  if (splits < 1) splits = 1;
  a.col_splits = (int)splits;
  a.tiles_per_split = (total_tiles + splits - 1) / splits;
  a.out = out;
  a.metric = metric;
  return launch_f32_t<MODE_DENSE, 16>(a, can_vec4(X, Y, d, dtype), rbs * splits, s);
}

int launch_sim_dense_combined(const float* F, const float* P, int64_t n, int64_t d, int64_t dp, float lambda_h,
                              float lambda_g, const float* nf, int64_t row0, int64_t rows, float* out, hipStream_t s) {
  if (n <= 0 || rows <= 0) return MMF_OK;
  ScanF32Args a{};
  a.X = F + row0 * d; a.Y = F; a.n = rows; a.m = n; a.d = d; a.dtype = MMF_F32;
  a.rx = nf + row0; a.cy = nf; a.row_ids = nullptr; a.n_rows = rows;
  a.neg_lambda = -lambda_h; a.kk = 0;
  const int64_t total_tiles = (n + F_CT - 1) / F_CT;
  const int64_t rbs = (rows + F_QT - 1) / F_QT;
  int64_t splits = (1024 + rbs - 1) / rbs;
  if (splits > total_tiles) splits = total_tiles;
  a.col_splits = (int)splits;
  a.tiles_per_split = (total_tiles + splits - 1) / splits;
  a.out = out; a.P = P; a.dp = (int)dp; a.neg_lambda_g = -lambda_g; a.prow0 = row0;
  a.metric = MMF_RBF;
  return launch_f32_t<MODE_DENSE, 16>(a, can_vec4(F, F, d, MMF_F32), rbs * splits, s);
}

}  // namespace mmf
