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
// Tiling: workgroup = 4 waves = 128 queries x 128 candidates per macro tile, K staged through LDS in chunks of
// 16 (scan) / 32 (dense), double buffered, one barrier per chunk.  Wave w owns queries 32w..32w+31 as the MFMA
// column (B operand) and sweeps the 4 candidate sub-tiles as MFMA rows (A operand), so every lane keeps ONE query
// and its candidate list is lane-private (mmf_dev.h).
//
// Operands come from the f32 IMAGES of X and Y (mmf_prep.hip: f32, zero padded to 128 rows / 32 columns, and inside
// every group of eight k de-interleaved to k0 k2 k4 k6 | k1 k3 k5 k7).  A 16-byte unit of the image is what one lane
// half of the MFMA feeds to four consecutive k-steps, so
//   * staging is LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction, 4 or 8 per wave and chunk): no
//     staging registers, no ds_write, no per-chunk address arithmetic (a loop-invariant lane offset + a scalar k offset);
//   * the chain reads ONE ds_read_b128 per operand per four k-steps (5 per 16 MFMAs), the next group's while the
//     current one multiplies.  LDS rows are KC floats, 16-byte units XOR-swizzled by row so that the 16 lanes of a
//     b128 access group hit 16 distinct bank quads.
// Round 2 measured the pieces in a bare loop (scripts/probe/f32_loop.hip, 2 workgroups per CU, of 157.3 TFLOP/s):
// register loop 0.98, + ds_read_b32 operands 0.95, + barrier 0.95, + staging through registers and ds_write_b32 (the
// round-1 kernel) 0.76, 16-byte [row][k] units with a per-half v_cndmask select 0.81 (VALU between the MFMAs is what
// costs), de-interleaved image + LDS-DMA + b128 reads 0.91.
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
// k per staged chunk (template parameter KC): 16 for the scan — 2 x (128 + 128) rows x 64 B = 32 KB beside 32 KB of
// lists, so two workgroups share a CU — and 32 for the dense outputs (no lists: two workgroups fit anyway, and half
// the barriers measure 7 % faster).

constexpr int MODE_SCAN = 0;
constexpr int MODE_DENSE = 1;

struct ScanF32Args {
  const float* Xp; const float* Yp;     // f32 images: row q of Xp is query position q of this launch
  int64_t n, m, dpad;
  const float* rx; const float* cy;
  const int32_t* row_ids; int64_t n_rows;
  float neg_lambda;
  int metric;
  int kk;
  int col_splits;
  int64_t tiles_per_split;
  uint32_t* cand_cnt; uint32_t* cand_ids; uint32_t* overflow;
  // k beyond one pass (k + self > 44): only columns that rank strictly AFTER (floor_key[q], floor_id[q]) in the total order
  // (key desc, id asc) are offered — the last entry the previous pass emitted for the row (NULL: no floor)
  const float* floor_key; const uint32_t* floor_id;
  // dense epilogue
  float* out;
  const float* P; int dp; float neg_lambda_g;
  int64_t prow0;   // dense combined, panel form: query q of this launch is row prow0 + q of P
  int debug;   // MMF_F32_DEBUG (timing-only ablations): 1 skip epilogue, 2 skip staging
};

template <int MODE, int CAP, int F_KC>
__global__ __launch_bounds__(F_NT, (MODE == MODE_DENSE || CAP <= 16) ? 2 : 1) void scan_f32_kernel(ScanF32Args a) {
  constexpr int UPR = F_KC / 4;            // 16-byte units per image row
  constexpr int RPP = 64 / UPR;            // image rows per 1 KiB DMA piece
  constexpr int HP = F_QT / RPP;           // pieces per operand tile (8 / 16)
  constexpr int PPW = 2 * HP / 4;          // pieces per wave: the first half of them query pieces, the rest candidate pieces
  constexpr int FSH = (UPR == 4) ? 2 : 1;  // unit u of row r sits at unit u ^ ((r >> FSH) & (UPR - 1))
  constexpr int NG = F_KC / 8;             // groups of eight k per chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Qs = reinterpret_cast<float*>(smem);   // [2][F_QT][F_KC]
  float* Cs = Qs + 2 * F_QT * F_KC;             // [2][F_CT][F_KC]
  float* cys = Cs + 2 * F_CT * F_KC;            // [2][F_CT] per-candidate scalars of the tile being accumulated
  float* lkeys = cys + 2 * F_CT;                // [CAP][F_NT]
  uint32_t* lids = reinterpret_cast<uint32_t*>(lkeys + CAP * F_NT);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  const int nkc = (int)(a.dpad / F_KC);
  const int64_t steps = (t_end - t_begin) * nkc;

  // this lane's query
  const int64_t qpos = q0 + 32 * wave + c;
  const bool qvalid = qpos < a.n_rows;
  const int64_t qrow = qvalid ? (a.row_ids ? (int64_t)a.row_ids[qpos] : qpos) : 0;
  const float ri = qvalid ? a.rx[qrow] : 1.0f;
  const int metric = a.metric;
  const bool floored = (MODE == MODE_SCAN) && a.floor_key != nullptr;
  const float fk = (floored && qvalid) ? a.floor_key[qpos] : 0.0f;
  const uint32_t fid = (floored && qvalid) ? a.floor_id[qpos] : 0u;

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

  // DMA roles: piece p = wave + 4 i covers image rows [(p % HP) * RPP, + RPP) of the query (p < HP) or candidate tile;
  // lane l lands at unit l of the piece = row l / UPR, unit l % UPR, and fetches the unit the swizzle puts there.
  // The lane's byte offset inside a tile image is loop invariant; chunk and tile move through scalar offsets.
  uint32_t voff[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int row = ((wave + 4 * i) % HP) * RPP + lane / UPR;
    const int lu = (lane % UPR) ^ ((row >> FSH) & (UPR - 1));
    voff[i] = (uint32_t)(((int64_t)row * a.dpad + 4 * lu) * 4);
  }
  const __amdgpu_buffer_rsrc_t qrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Xp + q0 * a.dpad), 0, -1, 0x00020000);
  float rcy = 0.0f;
  // chunk kc of candidate tile ct -> stage `buf`; the tile's 128 per-candidate scalars ride along with its first chunk
  auto stage = [&](int64_t ct, int kc, int buf) {
    if (kc == 0 && tid < F_CT) {
      int64_t j = ct * F_CT + tid;
      if (j > a.m - 1) j = a.m - 1;
      rcy = a.cy[j];
    }
    if (a.debug & 2) return;
    const __amdgpu_buffer_rsrc_t crsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Yp + ct * F_CT * a.dpad), 0, -1, 0x00020000);
    const int koff = kc * F_KC * 4;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pr = (wave + 4 * i) % HP;
      if (i < PPW / 2)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(qrsrc, (__attribute__((address_space(3))) void*)(Qs + buf * F_QT * F_KC + pr * RPP * F_KC),
                                                 16, (int)voff[i], koff, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, (__attribute__((address_space(3))) void*)(Cs + buf * F_CT * F_KC + pr * RPP * F_KC),
                                                 16, (int)voff[i], koff, 0, 0);
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  // this lane's operand rows inside a stage: query row 32 wave + c, candidate rows c + 32 t (same swizzle for all t)
  const int rqrow = 32 * wave + c;
  const int fq = (rqrow >> FSH) & (UPR - 1), fc = (c >> FSH) & (UPR - 1);

  if (steps > 0) {
    stage(t_begin, 0, 0);
    if (tid < F_CT) cys[tid] = rcy;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int64_t ct = t_begin;        // tile and chunk being multiplied
  int kc = 0;
  for (int64_t s = 0; s < steps; ++s) {
    const int buf = (int)(s & 1);
    int nkcn = kc + 1;
    int64_t nct = ct;
    if (nkcn == nkc) { nkcn = 0; nct = ct + 1; }
    if (s + 1 < steps) stage(nct, nkcn, buf ^ 1);

    const float* Qrow = Qs + buf * F_QT * F_KC + rqrow * F_KC;
    const float* Crow = Cs + buf * F_CT * F_KC + c * F_KC;
    // operands of group g + 1 are read from LDS before the MFMAs of group g issue
    f32x4 bn = *reinterpret_cast<const f32x4*>(Qrow + 4 * (half ^ fq));
    f32x4 avn[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) avn[t] = *reinterpret_cast<const f32x4*>(Crow + t * 32 * F_KC + 4 * (half ^ fc));
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const f32x4 b = bn;
      f32x4 av[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) av[t] = avn[t];
      if (g + 1 < NG) {
        bn = *reinterpret_cast<const f32x4*>(Qrow + 4 * ((2 * (g + 1) + half) ^ fq));
#pragma unroll
        for (int t = 0; t < 4; ++t) avn[t] = *reinterpret_cast<const f32x4*>(Crow + t * 32 * F_KC + 4 * ((2 * (g + 1) + half) ^ fc));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          // SCAN: C rows = candidates, columns = queries (a lane owns one query).  DENSE: the transpose — a lane owns
          // one candidate COLUMN of the output, so the 32 lanes of a half store 128 contiguous bytes of one output
          // row.  Products commute, the k order is the same: both orientations give the same bits.
          if constexpr (MODE == MODE_DENSE) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[e], av[t][e], acc[t], 0, 0, 0);
          else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t][e], b[e], acc[t], 0, 0, 0);
        }
      }
    }

    if (kc == nkc - 1 && !(a.debug & 1)) {
      const int tpar = (int)((ct - t_begin) & 1);
      // one 32 x 32 sub-tile at a time (keeping all four key vectors live costs 64 VGPRs and pushes the
      // staging registers of the next chunk into AGPRs, i.e. a vmcnt(0) in front of the MFMA block)
      if constexpr (MODE == MODE_DENSE) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int64_t j = ct * F_CT + 32 * t + c;            // this lane's output column
          const bool jv = j < a.m;
          const float cj = cys[tpar * F_CT + 32 * t + c];
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
        const float* cyt = cys + tpar * F_CT + 32 * t + 4 * half;
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
        if (floored) {                             // wave-uniform: a later pass of a large-k call
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const uint32_t j = (uint32_t)(cand0 + (r & 3) + 8 * (r >> 2) + 4 * half);
            const bool after = (key[r] < fk) || (key[r] == fk && j > fid);
            key[r] = after ? key[r] : kNegInf;
          }
        }
        if (__builtin_expect(__any(max16(key) >= list.thr), 0)) list.offer_tile(key, (uint32_t)cand0, half, a.kk);
      }
    }

      }
    if (s + 1 < steps && nkcn == 0 && tid < F_CT) cys[(int)((nct - t_begin) & 1) * F_CT + tid] = rcy;
    ct = nct; kc = nkcn;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of the next chunk have landed
    __syncthreads();
  }

  if constexpr (MODE == MODE_SCAN) {
    list.compact(a.kk);
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
  return sizeof(float) * (2 * F_QT * kc + 2 * F_CT * kc + 2 * F_CT) + (size_t)cap * F_NT * 8;
}

template <int MODE, int CAP>
static int launch_f32_t(const ScanF32Args& a, int64_t grid, hipStream_t s) {
  if (a.metric < MMF_DOT || a.metric > MMF_RBF) {
    set_error("scan_f32: unsupported metric %d", a.metric);
    return MMF_E_INVALID;
  }
  if (!a.Xp || !a.Yp || a.dpad <= 0 || (a.dpad & 31) != 0 || a.dpad > (int64_t(1) << 21)) {
    set_error("scan_f32: missing f32 operand images (or padded dim %lld not a multiple of 32 below 2^21)", (long long)a.dpad);
    return MMF_E_INTERNAL;
  }
  constexpr int KC = (MODE == MODE_SCAN) ? 16 : 32;
  const size_t lds = scan_f32_lds(MODE == MODE_SCAN ? CAP : 0, KC);
  auto kern = scan_f32_kernel<MODE, CAP, KC>;
  MMF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(F_NT), lds, s, a);
  MMF_LAUNCH_CHECK();
  return MMF_OK;
}

int launch_scan_f32(const ScanProblem& p, const CandLists& L, hipStream_t s, int* grid_out) {
  if (p.n_rows <= 0 || p.m <= 0) return MMF_OK;
  ScanF32Args a{};
  a.Xp = p.Xp; a.Yp = p.Yp; a.n = p.n; a.m = p.m; a.dpad = prep_f32_dim(p.d);
  a.rx = p.rx; a.cy = p.cy; a.row_ids = p.row_ids; a.n_rows = p.n_rows;
  a.neg_lambda = -p.lambda; a.kk = p.kk; a.col_splits = p.col_splits;
  const int64_t total_tiles = (p.m + F_CT - 1) / F_CT;
  a.tiles_per_split = (total_tiles + p.col_splits - 1) / p.col_splits;
  a.cand_cnt = L.cnt; a.cand_ids = L.ids; a.overflow = L.overflow;
  a.floor_key = p.floor_key; a.floor_id = p.floor_id;
  const int64_t grid = ((p.n_rows + F_QT - 1) / F_QT) * p.col_splits;
  if (grid_out) *grid_out = (int)grid;
  a.metric = p.metric;
  { const char* e = getenv("MMF_F32_DEBUG"); a.debug = e ? atoi(e) : 0; }
  if (L.cap == 16) return launch_f32_t<MODE_SCAN, 16>(a, grid, s);
  if (L.cap == 32) return launch_f32_t<MODE_SCAN, 32>(a, grid, s);
  if (L.cap == 48) return launch_f32_t<MODE_SCAN, 48>(a, grid, s);
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

bool sim_dense_needs_images(int64_t d, int metric) { return d > 8 && metric != MMF_RBF_DIRECT; }

int launch_sim_dense(const void* X, int64_t n, const void* Y, int64_t m, int64_t d, int dtype, int metric,
                     float lambda, const float* rx, const float* cy, const float* Xp, const float* Yp, float* out,
                     hipStream_t s) {
  if (n <= 0 || m <= 0) return MMF_OK;
  if (d <= 8 && metric != MMF_RBF_DIRECT) {
    dim3 grid((unsigned)((m + 1023) / 1024), (unsigned)((n + 15) / 16));
    hipLaunchKernelGGL(dense_small_d_kernel, grid, dim3(256), 0, s, X, n, Y, m, (int)d, dtype, metric, -lambda, rx, cy, out);
    MMF_LAUNCH_CHECK();
    return MMF_OK;
  }
  if (metric == MMF_RBF_DIRECT) return launch_rbf_direct(X, n, Y, m, d, dtype, lambda, out, nullptr, nullptr, s);   // mmf_direct.hip
  ScanF32Args a{};
  a.Xp = Xp; a.Yp = Yp; a.n = n; a.m = m; a.dpad = prep_f32_dim(d);
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
  { const char* e = getenv("MMF_F32_DEBUG"); a.debug = e ? atoi(e) : 0; }
  return launch_f32_t<MODE_DENSE, 16>(a, rbs * splits, s);
}

int launch_sim_dense_combined(const float* Fp, const float* P, int64_t n, int64_t d, int64_t dp, float lambda_h,
                              float lambda_g, const float* nf, int64_t row0, int64_t rows, float* out, hipStream_t s) {
  if (n <= 0 || rows <= 0) return MMF_OK;
  ScanF32Args a{};
  a.dpad = prep_f32_dim(d);
  a.Xp = Fp + row0 * a.dpad; a.Yp = Fp; a.n = rows; a.m = n;
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
  return launch_f32_t<MODE_DENSE, 16>(a, rbs * splits, s);
}

}  // namespace mmf
