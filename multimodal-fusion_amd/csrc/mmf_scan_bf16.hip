// mmf_scan_bf16.hip — the fast all-pairs scan: f16 / bf16 MFMA candidate generation with a proven
// error margin; the exact f32 re-rank (mmf_select.hip) turns the candidates into the final rows.
//
// Shape of the work: like a flash-attention S = Q K^T pass with head dim d and no
// softmax/PV.  A workgroup = 8 waves owns 256 queries for its whole column range (4 waves / 128 queries for d > 512):
//   * each wave keeps its 32 queries RESIDENT IN REGISTERS as the MFMA B operand
//     (KS x 8 halves = 128 VGPRs at d = 512), loaded once;
//   * candidate tiles of 32 rows x d stream L2/HBM -> LDS by buffer-form LDS-DMA (one 1 KiB piece per wave
//     instruction, scalar tile offset) into a 4-stage ring used as two pairs: an iteration reads one pair — TWO
//     tiles per s_barrier — while the pieces of the next pair are issued inside the MFMA chains; at the top of an
//     iteration a wave's outstanding DMA is exactly what the iteration needs (vmcnt(0), barrier);
//   * the LDS image is the candidates' natural row-major layout with the 16-byte chunk index XORed
//     by (row & 15) — applied on the DMA *source* address, LDS stays lane-linear — so every
//     ds_read_b128 of an A fragment is bank-conflict free;
//   * v_mfma_f32_16x16x32_{f16,bf16}, 2 x 2 tiles per wave (32 candidates x 32 queries), four independent
//     accumulator chains: C rows = candidates, C columns = queries, a lane holds TWO queries x 8 candidates per tile;
//     the accumulators are initialised with the per-candidate bias (-n_j/2 for the L2 metrics, -inf for padding)
//     instead of zero, which costs nothing;
//   * epilogue per tile: 14 v_max + 2 compares, reduced before the barrier, tested behind it; only a hit enters
//     the lane-private list code (SlotList, mmf_dev.h).
// Where the time goes (profiles/README.md, r02 ablations): the bare ds_read + MFMA chain alone runs at 0.60 of the
// 2.5 PFLOP/s peak (matrix pipe 73 % busy at the 1.94 GHz the chip holds under this load); the tile DMA adds 6 %
// (instruction issue, not traffic), the list code 10 %.
//
// Error margin: with u the f32 rows (normalised for cosine), z = round_16(u), the scanned value
// G = bias_j + z_i.z_j differs from the real-number target Q = bias_j + u_i.u_j by at most
//   E1_i = |dz_i| max|z_j| + |u_i| max|dz_j| + (DP+8) 2^-24 (|z_i| max|z_j| + max|bias|)
// and the canonical f32 key (mapped to Q units) differs from Q by at most E2_i (rounding of the
// chain and of the metric's few f32 ops).  Every column that can be in the canonical top-k has
// G >= (k-th best G) - 2(E1+E2); the lists keep exactly those, so the re-rank sees a superset.
// A list that cannot hold them hands the surplus to the row's overflow list; a row that fills that too is flagged
// and rescanned by the exact kernel.
#include <stdlib.h>

#include <type_traits>

#include "mmf_dev.h"
#include "mmf_host.h"

namespace mmf {

constexpr int B_CT = 32;
#ifndef MMF_B_CAP
#define MMF_B_CAP 15
#endif
constexpr int B_CAP = MMF_B_CAP;  // list entries per lane for k + self <= 11: what LDS leaves beside four 32 KiB tile stages
constexpr int B_CAP_BIG = 16;     // ... for k + self in 12..20 (costs the fourth tile stage: TPB = 1); what a lane list
                                  // cannot hold goes to the row's overflow list, so the capacity bounds speed, not k
constexpr int B_CAP_WIDE = 32;    // ... for k + self in 21..44, d <= 512: 5 slot bits, ranks counted out of LDS (64 KiB of lists
                                  // beside two 32 KiB tile stages)

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// prep: z = round(u) in bf16/f16, per-row norms of z, of the rounding residual and of u, bias,
// and the column-side maxima the margins need.  One wave per row.
// ------------------------------------------------------------------------------------------------
struct PrepArgs {
  const void* X; int64_t n; int64_t d; int dtype; int metric;
  const float* scal;        // canonical row scalars (n_i, or clamped norm for cosine)
  const uint32_t* max_n;    // float bits of the largest n over both operands (unused for cosine)
  void* Z; int64_t n_pad; int dp; int z_f16;
  float* zn; float* rn; float* un; float* cb;
  uint32_t* maxima;         // [4] float bits: max zn, max rn, max un, max |cb|
};

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

__global__ __launch_bounds__(256) void prep_half_kernel(PrepArgs a) {
  const int lane = threadIdx.x & 63;
  // Common exact power-of-two scale: the largest row norm lands in [256, 512), so f16 keeps its full
  // 11-bit precision on typical elements and cannot overflow.  Scaling by 2^e is exact, it multiplies
  // every scanned value by 2^(2e) and leaves the ranking untouched; the margins are computed from the
  // scaled norms, so they carry the same factor.
  float scale;
  {
    float mx = (a.metric == MMF_COSINE) ? 1.0f : __builtin_sqrtf(__uint_as_float(a.max_n[0]));
    int ex = 0;
    if (mx > 0.0f && mx < __builtin_huge_valf()) (void)frexpf(mx, &ex); else ex = 9;
    int e = 9 - ex;
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    scale = ldexpf(1.0f, e);
  }
  // one wave per row, grid-stride over rows; the four column-side maxima are kept per wave and
  // published once at the end (a same-address atomic per row would serialise the whole kernel)
  float m_zn = 0.f, m_rn = 0.f, m_un = 0.f, m_cb = 0.f;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t row = wave0; row < a.n_pad; row += nwaves) {
    uint16_t* zrow = reinterpret_cast<uint16_t*>(a.Z) + row * a.dp;
    if (row >= a.n) {  // padding rows: zeros, bias -inf so they can never be candidates
      for (int k = lane; k < a.dp; k += 64) zrow[k] = 0;
      if (lane == 0) { a.zn[row] = 0.f; a.rn[row] = 0.f; a.un[row] = 0.f; a.cb[row] = kNegInf; }
      continue;
    }
    const float sc = a.scal[row];
    float s_z = 0.f, s_r = 0.f, s_u = 0.f;
    const bool vec = (a.dtype == MMF_F32) && ((a.d & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.X) & 15) == 0);
    const bool vec16 = (a.dtype != MMF_F32) && ((a.d & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.X) & 7) == 0);
    for (int k4 = lane * 4; k4 < a.dp; k4 += 256) {       // 4 consecutive k per lane: 16 B in, 8 B out
      float u4[4] = {0.f, 0.f, 0.f, 0.f};
      if (vec) {
        if (k4 < a.d) {
          const f32x4 x4 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.X) + row * a.d + k4);
          u4[0] = x4[0]; u4[1] = x4[1]; u4[2] = x4[2]; u4[3] = x4[3];
        }
      } else if (vec16) {                                   // bf16 / f16 rows: 8 bytes per lane, exact upcasts
        if (k4 < a.d) {
          const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(a.X) + row * a.d + k4);
          if (a.dtype == MMF_BF16) {
            u4[0] = __uint_as_float(h.x << 16); u4[1] = __uint_as_float(h.x & 0xffff0000u);
            u4[2] = __uint_as_float(h.y << 16); u4[3] = __uint_as_float(h.y & 0xffff0000u);
          } else {
            u4[0] = (float)__builtin_bit_cast(_Float16, (uint16_t)(h.x & 0xffffu)); u4[1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(h.x >> 16));
            u4[2] = (float)__builtin_bit_cast(_Float16, (uint16_t)(h.y & 0xffffu)); u4[3] = (float)__builtin_bit_cast(_Float16, (uint16_t)(h.y >> 16));
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (k4 + i < a.d) u4[i] = ld_elem(a.X, row * a.d + k4 + i, a.dtype);
      }
      uint16_t b4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float u = u4[i];
        if (a.metric == MMF_COSINE) u = u / sc;
        u = u * scale;
        float z;
        if (a.z_f16) {
          const _Float16 hz = (_Float16)u;
          z = (float)hz;
          b4[i] = __builtin_bit_cast(uint16_t, hz);
        } else {
          b4[i] = f32_to_bf16_rne(u);
          z = bf16_bits_to_f32(b4[i]);
        }
        const float r = z - u;
        s_z = __builtin_fmaf(z, z, s_z);
        s_r = __builtin_fmaf(r, r, s_r);
        s_u = __builtin_fmaf(u, u, s_u);
      }
      uint2 pk;
      pk.x = (uint32_t)b4[0] | ((uint32_t)b4[1] << 16);
      pk.y = (uint32_t)b4[2] | ((uint32_t)b4[3] << 16);
      *reinterpret_cast<uint2*>(zrow + k4) = pk;            // dp is a multiple of 128: always in range
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s_z += __shfl_xor(s_z, o);
      s_r += __shfl_xor(s_r, o);
      s_u += __shfl_xor(s_u, o);
    }
    const float up = 1.0f + 1e-4f;  // covers the rounding of these sums and square roots
    const float zn = __builtin_sqrtf(s_z) * up, rn = __builtin_sqrtf(s_r) * up, un = __builtin_sqrtf(s_u) * up;
    const float cb = (a.metric == MMF_NEG_SQ_L2 || a.metric == MMF_RBF) ? (-0.5f * sc * scale * scale) : 0.0f;
    if (lane == 0) { a.zn[row] = zn; a.rn[row] = rn; a.un[row] = un; a.cb[row] = cb; }
    m_zn = fmaxf(m_zn, zn); m_rn = fmaxf(m_rn, rn); m_un = fmaxf(m_un, un); m_cb = fmaxf(m_cb, fabsf(cb));
  }
  // Publish once per WORKGROUP: same-address atomics retire at about 90 per microsecond, so one set per wave
  // (16384 x 4 of them) used to be most of this kernel's time.  Non-negative floats order like their bits.
  __shared__ float wmax[4][4];
  const int w = threadIdx.x >> 6;
  if (lane == 0) { wmax[w][0] = m_zn; wmax[w][1] = m_rn; wmax[w][2] = m_un; wmax[w][3] = m_cb; }
  __syncthreads();
  if (threadIdx.x < 4) {
    const float m = fmaxf(fmaxf(wmax[0][threadIdx.x], wmax[1][threadIdx.x]), fmaxf(wmax[2][threadIdx.x], wmax[3][threadIdx.x]));
    if (m > 0.f) atomicMax(a.maxima + threadIdx.x, __float_as_uint(m));
  }
}

// ------------------------------------------------------------------------------------------------
// scan
// ------------------------------------------------------------------------------------------------
struct ScanB16Args {
  const void* ZQ;            // [nq_pad][DP] query side
  const void* ZC;            // [m_pad][DP]  candidate side
  const float* cb;           // [m_pad]
  const float* q_zn; const float* q_rn; const float* q_un;
  const uint32_t* maxima;    // candidate-side maxima
  int64_t n_rows;            // queries
  int64_t m;                 // real candidates
  int64_t tiles_total;       // m_pad / 32
  int64_t tiles_per_split;
  int64_t row_blocks;
  int col_splits;            // power of two
  int conc_splits;           // splits that run side by side (power of two <= 8); the rest follow in later "rounds" of block ids
  int64_t blocks_per_round;
  int lists_total;           // lists per query row in cand_cnt / cand_ids (>= list_base + 2 * col_splits)
  int list_base;             // first list slot written by this launch
  uint32_t seg_len, seg_stride, id_off;   // column i of ZC is reported as id_off + (i / seg_len) * seg_stride + i % seg_len
                                          // (seg_len == 0: id_off + i)
  int32_t* seed;             // [row_blocks * QT] best threshold published for each query by the workgroups / launches
                             // that scan it (order-preserving int encoding, see seed_enc; memset 0x80 = none)
  int32_t* lost;             // [row_blocks * QT] best key any of them dropped (same encoding); the row is rescanned
                             // exactly iff lost >= seed once every launch has finished (audit_kernel)
  int share;                 // other workgroups scan the same queries: import their thresholds on the way
  int kk;
  int metric;
  int d;
  int debug;                 // MMF_SCAN_DEBUG (instrumented build of the same template; scripts/ablate.py): 1 = skip the filter /
                             // list code, 2 = no tile DMA, 4 = no barrier, 32 = DMA of tile 0 only (1, 2, 4, 32: timing only, the
                             // results are wrong), 8 = count events, 16 = cycle stamps, 64 = instrumented build, nothing removed
  unsigned long long* dbg;   // [8] event counters when debug & 8
  uint32_t* lids;            // lane-private id slots: [grid][16][B_NT] (global, written on push, read once at the end)
  uint32_t* cand_cnt; uint32_t* cand_ids; uint32_t* overflow;
  float* cand_keys;          // approximate keys of the entries (select prunes with them), or nullptr
  float* margin_out;         // [n_rows] the queries' error margins, written with cand_keys
  uint32_t* spill_cnt; uint32_t* spill_ids; int spill_cap;   // per-row overflow lists (SpillSink), or nullptr / 0
  int spill_stacks;          // one list pair per row: the two lanes fill the row's slots from both ends, no counter (SpillSink)
};

// Order-preserving float <-> int32 map (an involution) so that thresholds can be merged with atomicMax.
__device__ __forceinline__ int32_t seed_enc(float f) {
  const int32_t b = __float_as_int(f);
  return b >= 0 ? b : (b ^ 0x7fffffff);
}
constexpr int32_t kSeedNone = (int32_t)0x80808080;   // hipMemset(0x80) pattern: "no threshold yet"

// LDS-DMA (global -> LDS, no registers) of the 32 biases of a tile.  LDS address = wave-uniform base + lane * 4.
__device__ __forceinline__ void glds4(const void* gptr, const void* lptr) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(lptr), 4, 0, 0);
}

// MFMA shape: v_mfma_f32_16x16x32_{f16,bf16}.  The kernel was first written on the 32x32x16 shape (one 32 x 32 tile
// per wave, a lane = one query x 16 candidates); operand bytes, LDS image and cycles per flop are identical, but the
// chip holds a higher clock on 16x16x32 under load (MI355X_MICROARCH.md "DVFS give-back" item 7): 10 % less wall
// time on the same box (profiles/README.md; the 32x32x16 form is in the history, commit "scan on v_mfma_f32_16x16x32").
// C layout per 16x16 tile: lane l -> column l & 15, rows 4 (l >> 4) + {0..3}.  A wave covers its 32 queries x the
// 32 candidates of a tile with 2 x 2 tiles acc[cb][qb]: a lane holds TWO queries (l & 15, + 16) x 8 candidates.
// The list code keeps its "one lane = one query, 16 candidates" form: lane l owns query (l & 15) + 16 ((l >> 4) & 1);
// on the (rare) slow path the lanes l and l ^ 16 swap the 8 values they hold for each other's query.
// NW waves per workgroup (32 queries each), TPB tiles per barrier (2 * TPB tile stages in LDS):
//   d <= 512 : NW = 8 (two waves per SIMD, 256 VGPRs each), TPB = 2
//   d <= 1024: SPLITK — NW = 8, TPB = 1: the waves w and w + 4 share 32 queries and a SIMD; each keeps HALF of every
//              query's k range resident (128 VGPRs), walks that half of every tile, and the upper-half wave hands its
//              partial accumulators to the lower-half wave through LDS (one 16 KiB exchange block, acknowledged per
//              tile), which adds them behind the next barrier and runs the filter / lists.  Two waves per SIMD hide
//              each other's DMA issue, as at d <= 512.
//              (k + self in 12..20 at d <= 1024: NW = 4, one wave per SIMD with all of k — its 16-entry lists leave no
//              room for the exchange block.)
template <int KS, bool F16, bool DBG, int NW, int TPB, int CAP, bool SPLITK = false>
__global__ __launch_bounds__(64 * NW, (NW == 8) ? 2 : 1) void scan_b16x_kernel(ScanB16Args a) {
  static_assert(!SPLITK || (TPB == 1 && NW == 8 && (KS % 8) == 0), "split-k pairs");
  constexpr int NQW = SPLITK ? NW / 2 : NW;     // waves that own queries and lists
  constexpr int NTL = 64 * NQW;                 // threads that own lists
  constexpr int QT = 32 * NQW;
  constexpr int STAGES = 2 * TPB;
  constexpr int ROWB = KS * 32;                 // bytes per candidate row in LDS (= DP * 2)
  constexpr int TILEB = B_CT * ROWB;            // bytes per tile
  constexpr int PIECES = TILEB / 1024;          // 1 KiB DMA pieces per tile (= KS)
  constexpr int PPW = (PIECES + NW - 1) / NW;
  static_assert(PIECES % NW == 0, "piece distribution");
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* tiles = smem;                                                  // [STAGES][TILEB]
  float* cbs = reinterpret_cast<float*>(smem + STAGES * TILEB);      // [STAGES][64]
  // Candidate lists (SlotList, mmf_dev.h): approximate keys in LDS, column ids in a global slot block of
  // this workgroup — leaving most of LDS to the tile ring is what buys the fourth stage.
  float* lkeys = cbs + STAGES * 64;                                  // [CAP][NTL]
  uint32_t* lids = a.lids + (size_t)blockIdx.x * (SlotList<CAP, NTL>::SLOTS * NTL);
  f32x4* xch = reinterpret_cast<f32x4*>(lkeys + CAP * NTL);          // SPLITK: [NQW][4][64] partial accumulators
  volatile int* ack = reinterpret_cast<volatile int*>(xch + NQW * 4 * 64);   // SPLITK: [NQW] last tile the lower wave took

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int khalf = SPLITK ? wave / NQW : 0;    // which half of k this wave multiplies
  const int qw = SPLITK ? wave % NQW : wave;    // which 32 queries
  const bool owner = khalf == 0;                // bias, filter and lists live in the lower-half wave
  const int half = lane >> 5;
  const int g = lane >> 4;                      // row group of the C layout: rows 4 g .. 4 g + 3 of each 16-row tile
  const int c16 = lane & 15;
  const int ownb = g & 1;                       // the query block (0 / 1) whose list this lane owns
  const int c = c16 + 16 * ownb;                // own query within the wave

  // XCD-aware block -> (row block, split): blocks that share blockIdx % 8 share an XCD (observed
  // round-robin placement; speed only) and are given the same column range, so its tiles are L2 hits.
  // Splits beyond the `conc_splits` that run side by side follow in later rounds of block ids, i.e. later in
  // dispatch order: their workgroups start from the thresholds the earlier ones published (a.seed).
  int64_t rb;
  int split;
  {
    const int CS = a.conc_splits;
    const int64_t round = blockIdx.x / a.blocks_per_round;
    const int64_t j = blockIdx.x - round * a.blocks_per_round;
    const int64_t q = j >> 3;
    const int x = (int)(j & 7);
    split = (int)round * CS + x % CS;
    rb = q * (8 / CS) + x / CS;
  }
  if (rb >= a.row_blocks) return;
  const int64_t q0 = rb * QT;
  int64_t t_begin = (int64_t)split * a.tiles_per_split;
  int64_t t_end = t_begin + a.tiles_per_split;
  if (t_end > a.tiles_total) t_end = a.tiles_total;
  if (t_begin > t_end) t_begin = t_end;
  const int64_t T = t_end - t_begin;

  const int64_t qpos = q0 + 32 * qw + c;
  const bool qvalid = (qpos < a.n_rows) && owner;

  // margin of this lane's query (see the header): 2 (E1 + E2)
  float margin;
  {
    const float ZB = __uint_as_float(a.maxima[0]), RB = __uint_as_float(a.maxima[1]);
    const float UB = __uint_as_float(a.maxima[2]), CB = __uint_as_float(a.maxima[3]);
    const float zn = a.q_zn[qpos], rn = a.q_rn[qpos], un = a.q_un[qpos];   // arrays are padded
    const float g_acc = (float)(KS * 16 + 8) * 5.9604645e-8f;
    const float g_chain = (float)(a.d + 2) * 5.9604645e-8f;
    // + 2^-19 |G| (2^-18 with 5 slot bits): the slot number a stored key carries in its low mantissa bits (SlotList)
    const float slot_eps = (CAP <= 16) ? 1.9073486e-6f : 3.8146973e-6f;
    const float e1 = rn * ZB + un * RB + (g_acc + slot_eps) * (zn * ZB + CB);
    float e2;
    if (a.metric == MMF_DOT) e2 = g_chain * un * UB;
    else if (a.metric == MMF_COSINE) e2 = (g_chain + 4.7683716e-7f) * un * UB * 1.01f;
    else e2 = g_chain * un * UB + 2.3841858e-7f * (un * un + UB * UB);
    margin = 2.0f * (e1 + e2) * 1.001f + 1e-30f;
  }

  SlotList<CAP, NTL> list;
  list.init(lkeys + (tid & (NTL - 1)), lids + (tid & (NTL - 1)));
  if (!qvalid) list.thr = __builtin_huge_valf();
  else if (a.spill_cnt) {
    list.sink.cnt = a.spill_cnt; list.sink.ids = a.spill_ids; list.sink.cap = (uint32_t)a.spill_cap; list.sink.row = qpos;
    list.sink.seg_len = a.seg_len; list.sink.seg_stride = a.seg_stride; list.sink.id_off = a.id_off;
    if (DBG) list.sink.ablate = (a.debug & 128) ? 1 : 0;
    if (a.spill_stacks) list.sink.stacks = 1 + half;
  }
  // Thresholds are shared between the workgroups (and launches) that scan different columns for the same
  // queries: any list's threshold bounds the approximate key of every member of the final top-k, whatever
  // columns it sits in.  sync_seed publishes this lane's threshold when it has risen (and is not the product
  // of a dropped key) and adopts the best one published so far; it runs at the start and every 64 tiles.
  float pub = -kFltMax;
  auto sync_seed = [&]() {
    if (!qvalid) return;
    if (list.thr > pub && list.thr > list.lost) {
      pub = list.thr;
      atomicMax(a.seed + qpos, seed_enc(pub));
    }
    if (!a.share) return;
    const int32_t o = __hip_atomic_load(a.seed + qpos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (o > kSeedNone) {
      const float t = __int_as_float(o >= 0 ? o : (o ^ 0x7fffffff));
      if (t > list.thr) { list.thr = t; pub = t; }
    }
  };
  if (a.share) sync_seed();
  // a wave whose queries all start from a published threshold skips the cold-start treatment
  const bool seeded = !__any(list.thr == -kFltMax);

  // resident query fragments: B operand of k-step s (32 wide), query block qb: lane holds
  // Z[q0 + 32 w + 16 qb + c16][32 s + 8 g .. +7]
  constexpr int KS2 = SPLITK ? KS / 4 : KS / 2;      // 32-wide k-steps this wave multiplies
  const int kbyte = khalf * KS2 * 64;                // byte offset of its k range inside a row
  u32x4 qf[KS2][2];
  {
    const char* qbase = reinterpret_cast<const char*>(a.ZQ) + (q0 + 32 * qw + c16) * (int64_t)ROWB + g * 16 + kbyte;
#pragma unroll
    for (int s = 0; s < KS2; ++s) {
      qf[s][0] = *reinterpret_cast<const u32x4*>(qbase + s * 64);
      qf[s][1] = *reinterpret_cast<const u32x4*>(qbase + 16 * (int64_t)ROWB + s * 64);
    }
    // make the compiler retire these loads HERE: otherwise it cannot prove them complete at their first use
    // inside the tile loop and puts s_waitcnt vmcnt(0) there, which drains the tile DMA in flight every iteration
#pragma unroll
    for (int s = 0; s < KS2; ++s) { asm volatile("" : "+v"(qf[s][0])); asm volatile("" : "+v"(qf[s][1])); }
  }

  // A-fragment read offsets inside a tile: row c16 (+ 16 cb), 16-byte chunk (4 s + g) ^ (row & 15)
  int lo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) lo[j] = c16 * ROWB + ((((4 * j + g) ^ c16) & 15) << 4);

  // DMA roles: piece p of a tile covers LDS bytes [1024 p, 1024 p + 1024); lane writes 16 B at l*16.
  // Source = wave-uniform running tile pointer + a loop-invariant 32-bit lane offset.
  uint32_t src_off[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int p = wave + NW * i;
    const int off = p * 1024 + lane * 16;
    const int r = off / ROWB;
    const int chunk = (off % ROWB) >> 4;
    const int src_chunk = (chunk & ~15) | ((chunk ^ r) & 15);
    src_off[i] = (uint32_t)(r * ROWB + src_chunk * 16);
  }
  const char* zc0 = reinterpret_cast<const char*>(a.ZC) + t_begin * (int64_t)TILEB;          // tile 0 of my range
  const char* cb0 = reinterpret_cast<const char*>(a.cb + t_begin * B_CT);
  // Tile pieces go through the buffer form of the LDS-DMA: the workgroup's column range is one raw buffer
  // (base = its first tile), the running tile position is the scalar offset and the lane's swizzled position
  // the vector offset — no per-piece 64-bit address arithmetic in the MFMA chain, and (unlike the FLAT-encoded
  // global form) it leaves the compiler's counted lgkmcnt waits for the A-fragment ring intact.
  const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(zc0), 0, -1, 0x00020000);
  auto issue_piece = [&](const char* tsrc, int stage, int i) {
    if (DBG && (a.debug & 2)) return;       // timing-only ablation: no tile DMA (the chain runs on whatever LDS holds)
    if (DBG && (a.debug & 32)) tsrc = zc0;  // timing-only ablation: every piece re-fetches tile 0 (issue cost without the traffic)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(zrsrc, (__attribute__((address_space(3))) void*)(tiles + stage * TILEB + (wave + NW * i) * 1024),
                                             16, (int)src_off[i], (int)(uint32_t)(tsrc - zc0), 0, 0);
  };
  // the 32 biases of a tile: one 4-byte DMA by lanes 0..31 of the wave whose turn it is
  // (buffer form like the tile pieces: the FLAT-encoded global form makes the compiler wait lgkmcnt(0) where the
  // A-fragment ring wants counted waits)
  const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(cb0), 0, -1, 0x00020000);
  auto issue_bias = [&](const char* bsrc, int stage) {
    const uint32_t l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // lane id, no live VGPR
    if (l < 32)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(brsrc, (__attribute__((address_space(3))) void*)(cbs + stage * 64), 4, (int)(l * 4u),
                                               (int)(uint32_t)(bsrc - cb0), 0, 0);
  };

  if (SPLITK && tid < NQW) ack[tid] = -1;
  const int Ti = (int)T;
  if (Ti > 0) {   // first group of TPB tiles -> stages 0 .. TPB-1
#pragma unroll
    for (int u = 0; u < TPB; ++u) {
      const bool real = u < Ti;
#pragma unroll
      for (int i = 0; i < PPW; ++i) issue_piece(real ? zc0 + u * (int64_t)TILEB : zc0, u, i);
      if (wave == (u & (NW - 1))) issue_bias(real ? cb0 + u * B_CT * 4 : cb0, u);
    }
  }
  // The DMA pieces of tile t+TPB are issued inside the MFMA chain of tile t, one per group of GRP MFMAs.  The
  // chain is straight-line code and the DMA is unconditional — past the end of the range it re-fetches tile 0
  // into a stage nobody reads again — so no branch and no per-tile bookkeeping sits between the MFMAs.
  // (Issuing at different points of the chain for the two waves that share a SIMD used to pay when a piece cost
  // 64-bit address arithmetic; with the buffer form it measures 0.8 % slower than issuing at the same point.)
  constexpr int GRP = KS2 / PPW;                     // k-steps (4 MFMAs each) between two DMA pieces (issuing the SPLITK
                                                     // kernel's pieces in its first k-steps instead measured the same)
  static_assert(KS2 % PPW == 0 && GRP >= 1, "DMA piece placement");
  const char* tsrc = zc0 + TPB * (int64_t)TILEB;     // source of the first tile of the NEXT group
  const char* bsrc = cb0 + TPB * B_CT * 4;
  const uint32_t id_base = (uint32_t)(t_begin * B_CT);

  struct Acc { f32x4 t[2][2]; };                     // [candidate block][query block]
  auto tile_body = [&](const char* tb, const float* cbt, const char* src, int s2) -> Acc {
    Acc acc;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      f32x4 b4 = *reinterpret_cast<const f32x4*>(cbt + 16 * cb + 4 * g);
      if (SPLITK && !owner) b4 = f32x4{0.f, 0.f, 0.f, 0.f};      // the bias is added once, by the lower-half wave
      acc.t[cb][0] = b4; acc.t[cb][1] = b4;
    }
    // A fragments are read PD k-steps ahead of the MFMAs that consume them (explicit register ring)
    constexpr int PD = 2;
    u32x4 af[PD][2];
#pragma unroll
    for (int s = 0; s < PD; ++s) {
      af[s][0] = *reinterpret_cast<const u32x4*>(tb + lo[s & 3] + (s >> 2) * 256);
      af[s][1] = *reinterpret_cast<const u32x4*>(tb + lo[s & 3] + (s >> 2) * 256 + 16 * ROWB);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * PD, 0);   // the leading reads first
#pragma unroll
    for (int s = 0; s < KS2; ++s) {
      const u32x4 a0 = af[s % PD][0], a1 = af[s % PD][1];
      if (s + PD < KS2) {
        af[s % PD][0] = *reinterpret_cast<const u32x4*>(tb + lo[(s + PD) & 3] + ((s + PD) >> 2) * 256);
        af[s % PD][1] = *reinterpret_cast<const u32x4*>(tb + lo[(s + PD) & 3] + ((s + PD) >> 2) * 256 + 16 * ROWB);
      }
      if constexpr (F16) {
        acc.t[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a0), __builtin_bit_cast(f16x8_t, qf[s][0]), acc.t[0][0], 0, 0, 0);
        acc.t[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a0), __builtin_bit_cast(f16x8_t, qf[s][1]), acc.t[0][1], 0, 0, 0);
        acc.t[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a1), __builtin_bit_cast(f16x8_t, qf[s][0]), acc.t[1][0], 0, 0, 0);
        acc.t[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a1), __builtin_bit_cast(f16x8_t, qf[s][1]), acc.t[1][1], 0, 0, 0);
      } else {
        acc.t[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a0), __builtin_bit_cast(bf16x8_t, qf[s][0]), acc.t[0][0], 0, 0, 0);
        acc.t[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a0), __builtin_bit_cast(bf16x8_t, qf[s][1]), acc.t[0][1], 0, 0, 0);
        acc.t[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a1), __builtin_bit_cast(bf16x8_t, qf[s][0]), acc.t[1][0], 0, 0, 0);
        acc.t[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a1), __builtin_bit_cast(bf16x8_t, qf[s][1]), acc.t[1][1], 0, 0, 0);
      }
      // one DMA piece of tile t+TPB per GRP k-steps
      if ((s % GRP) == GRP - 1) issue_piece(src, s2, s / GRP);
      // pin the interleave: the two LDS reads of step s+PD, then the four MFMAs of step s
      if (s + PD < KS2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
    return acc;
  };
  // per query block: the largest of the 8 values this lane holds for it
  auto max_qb = [&](const Acc& x, int qb) -> float {
    const f32x4& p = x.t[0][qb];
    const f32x4& q = x.t[1][qb];
    return fmaxf(fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3])), fmaxf(fmaxf(q[0], q[1]), fmaxf(q[2], q[3])));
  };
  // thresholds of the two queries this lane holds values for: its own list's and its partner's (lane ^ 16);
  // they change only inside the list code and in sync_seed, and are re-read from the partner there
  float thr_q0, thr_q1;
  auto refresh_thr = [&]() {
    const float mine = list.thr;
    const float theirs = __shfl_xor(mine, 16);
    thr_q0 = ownb ? theirs : mine;
    thr_q1 = ownb ? mine : theirs;
  };
  refresh_thr();

  // The filter of tile t-1 runs AFTER barrier t, ahead of this wave's own MFMA chain: a wave that
  // falls into the (rare) list code then delays only itself while its SIMD partner issues MFMAs;
  // placed before the barrier it would hold all eight waves, and their matrix pipes, at the barrier.
  auto filter = [&](const Acc& acc, int tt, float m0, float m1) {
    if (__builtin_expect(!(DBG && (a.debug & 1)) && __any((m0 >= thr_q0) || (m1 >= thr_q1)), 0)) {
      const uint32_t id0 = id_base + (uint32_t)tt * B_CT;
      // this lane's query gets its 16 candidates together: its own 8 plus the 8 the partner lane holds
      f32x16 v;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float mine = ownb ? acc.t[cb][1][j] : acc.t[cb][0][j];
          const float give = ownb ? acc.t[cb][0][j] : acc.t[cb][1][j];
          v[4 * cb + j] = mine;
          v[8 + 4 * cb + j] = __shfl_xor(give, 16);
        }
      }
      // v[0..7]: rows of this lane's group g, v[8..15]: rows of the partner's group g ^ 1 (4 g ^ 4)
      const int g4 = 4 * g;
      auto rowof = [g4](int r) -> uint32_t { return (uint32_t)((g4 ^ ((r & 8) >> 1)) + (r & 3) + 16 * ((r >> 2) & 1)); };
      // robust (never dropping) path while thresholds are still forming: first 32 tiles of the range
      const bool cold = (!seeded && tt < 32) || __any(list.thr == -kFltMax);
      if (DBG && (a.debug & 8)) {
        const bool willc = __any(list.cnt >= CAP - 1);
        int nh = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) nh += (v[r] >= list.thr) ? 1 : 0;
        const int lanes_hit = __popcll(__ballot(nh > 0));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nh += __shfl_xor(nh, o);
        if (lane == 0) {
          atomicAdd(a.dbg + 0, 1ull);                       // slow-path entries
          atomicAdd(a.dbg + 1, cold ? 1ull : 0ull);         // ... of which cold
          atomicAdd(a.dbg + 2, (unsigned long long)nh);     // hits (values >= thr)
          atomicAdd(a.dbg + 3, (!cold && willc) ? 1ull : 0ull);   // warm compactions
          atomicAdd(a.dbg + 4, (unsigned long long)lanes_hit);
        }
      }
      unsigned long long ts0 = 0;
      if (DBG && (a.debug & 16)) ts0 = __builtin_amdgcn_s_memtime();
      list.offer_tile(v, id0, rowof, a.kk, margin);
      if (DBG && (a.debug & 16) && lane == 0) {
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - ts0;
        atomicAdd(a.dbg + (cold ? 4 : 5), dt);
        atomicAdd(a.dbg + (cold ? 6 : 7), 1ull);
      }
      refresh_thr();
    }
    if (DBG && (a.debug & 8) && lane == 0) atomicAdd(a.dbg + 5, 1ull);   // tiles
    if (__builtin_expect(a.share && (tt & 63) == 63, 0)) { sync_seed(); refresh_thr(); }
  };

  // Main loop: TPB tiles per barrier.  Iteration j reads the group of stages holding tiles j*TPB ..
  // and fills the other group with the next TPB tiles (DMA pieces issued inside the MFMA chains), so at
  // the top of an iteration everything this wave has in flight is exactly what the iteration needs:
  // vmcnt(0), barrier.  Past the end of the range the DMA re-fetches tile 0 into a stage nobody reads.
  unsigned long long tw = 0, tf = 0, tc = 0, tv = 0, t0s = 0, t1s = 0, t2s = 0, t3s = 0, tvs = 0;
  const bool stamps = DBG && (a.debug & 16) != 0;
  Acc acc_prev;
  float m0_prev = kNegInf, m1_prev = kNegInf;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc_prev.t[cb][0][j] = kNegInf; acc_prev.t[cb][1][j] = kNegInf; }   // "tile -1"
  }
  const int nIter = (Ti + TPB - 1) / TPB;
  for (int j = 0; j < nIter; ++j) {
    if (stamps) tvs = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) alone (expcnt 7, lgkmcnt 15 = no wait)
    if (stamps) t0s = __builtin_amdgcn_s_memtime();
    asm volatile("" ::: "memory");
    if (!(DBG && (a.debug & 4)))    // timing-only ablation: no workgroup barrier (races)
    __builtin_amdgcn_s_barrier();   // this group is visible to all; everyone is done READING the other group
    asm volatile("" ::: "memory");
    if (stamps) t1s = __builtin_amdgcn_s_memtime();

    if constexpr (SPLITK) {
      if (owner) {
        if (j > 0) {                 // the partner's half of tile j - 1, written before the barrier just passed
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f32x4 p = xch[(qw * 4 + i) * 64 + lane];
            acc_prev.t[i >> 1][i & 1] += p;
          }
          m0_prev = max_qb(acc_prev, 0);
          m1_prev = max_qb(acc_prev, 1);
        }
        // the reads of the exchange block above must be ISSUED before the acknowledgement below (LDS operations of a wave
        // complete in order, so issue order is all it takes): keep the compiler from sinking them past the volatile store
        asm volatile("" ::: "memory");
        if (lane == 0) ack[qw] = j;  // the exchange block may be overwritten
      }
    }
    if (owner) filter(acc_prev, j * TPB - 1, m0_prev, m1_prev);  // behind the barrier: only this wave waits for its own list code
    if (stamps) t2s = __builtin_amdgcn_s_memtime();

    const int sg = (j & 1) * TPB, ng = TPB - sg;      // stage group read / filled by this iteration
#pragma unroll
    for (int u = 0; u < TPB; ++u) {
      const int t = j * TPB + u;
      const bool more = (t + TPB < Ti);
      const char* src = more ? tsrc + u * (int64_t)TILEB : zc0;
      // the 32 biases of tile t + TPB go out BEFORE this tile's chain: issued behind it, the wave whose turn it is would
      // reach the vmcnt(0) of the next iteration with a DMA just issued, and seven waves would wait for it at the barrier
      Acc acc;
      acc = tile_body(tiles + (sg + u) * TILEB + kbyte, cbs + (sg + u) * 64, src, ng + u);
      // (behind the chain: issued in front of it, the same DMA costs +8 %; mid-chain needs a branch inside the chain)
      if (wave == ((t + TPB) & (NW - 1))) issue_bias(more ? bsrc + u * B_CT * 4 : cb0, ng + u);
      if (u < TPB - 1) {
        if (TPB == 2 || __builtin_expect(t < Ti, 1))      // (TPB > 2: the ragged last group can hold several dummy tiles)
        filter(acc, t, max_qb(acc, 0), max_qb(acc, 1));   // mid-iteration, no barrier nearby
      } else {
        if (__builtin_expect(t >= Ti, 0)) {              // ragged range: the last tile of the last group is a dummy
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { acc.t[cb][0][jj] = kNegInf; acc.t[cb][1][jj] = kNegInf; }
          }
        }
        acc_prev = acc;
        if constexpr (!SPLITK) {
          m0_prev = max_qb(acc, 0);     // reduced BEFORE the barrier, in time this wave would otherwise spend waiting
          m1_prev = max_qb(acc, 1);
        } else if (!owner) {
          // hand this tile's partial sums to the lower-half wave once it has taken the previous tile's
          while (ack[qw] < j) __builtin_amdgcn_s_sleep(1);
          asm volatile("" ::: "memory");        // ... and the writes of the next partials stay behind the poll
#pragma unroll
          for (int i = 0; i < 4; ++i) xch[(qw * 4 + i) * 64 + lane] = acc.t[i >> 1][i & 1];
          __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the partials are in LDS before this wave reaches the barrier
        }
      }
    }
    tsrc += TPB * (int64_t)TILEB;
    bsrc += TPB * B_CT * 4;
    if (stamps) {
      asm volatile("" :: "v"(acc_prev.t[0][0][0]));  // the chain's result must exist before the stamp
      t3s = __builtin_amdgcn_s_memtime();
      tw += t1s - t0s; tf += t2s - t1s; tc += t3s - t2s; tv += t0s - tvs;
    }
  }
  if (stamps && lane == 0) {
    atomicAdd(a.dbg + 0, tw); atomicAdd(a.dbg + 1, tf); atomicAdd(a.dbg + 2, tc); atomicAdd(a.dbg + 3, (unsigned long long)Ti);
    atomicAdd(a.dbg + 8, tv);
  }
  if constexpr (SPLITK) {               // the last tile's upper half
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
    if (owner && Ti > 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc_prev.t[i >> 1][i & 1] += xch[(qw * 4 + i) * 64 + lane];
    }
  }
  if (Ti > 0 && owner) filter(acc_prev, nIter * TPB - 1, max_qb(acc_prev, 0), max_qb(acc_prev, 1));
  __builtin_amdgcn_s_waitcnt(0x0F70);   // the dummy tiles still in flight

  list.compact(a.kk, margin);
  list.sink.close();                    // reserved overflow-list slots this lane did not use
  if (a.spill_stacks && a.spill_cnt) {  // two-stack overflow lists: both counts in one word; stacks that met cost the row its fast path
    const uint32_t mine = qvalid ? list.sink.cpos : 0u;
    const uint32_t theirs = (uint32_t)__shfl_xor((int)mine, 32);
    if (mine + theirs > (uint32_t)a.spill_cap) list.lost = kFltMax;
    if (qvalid && half == 0 && (mine | theirs)) a.spill_cnt[qpos] = mine | (theirs << 16);
  }
  sync_seed();
  if (qvalid) {
    const int64_t lbase = qpos * a.lists_total + a.list_base + 2 * split + half;
    a.cand_cnt[lbase] = (uint32_t)list.cnt;
    for (int e = 0; e < list.cnt; ++e) {
      uint32_t id = list.id_of(e);
      if (a.seg_len) id = (id / a.seg_len) * a.seg_stride + id % a.seg_len;
      a.cand_ids[lbase * CAP + e] = id + a.id_off;
      if (a.cand_keys) a.cand_keys[lbase * CAP + e] = list.keys[e * NTL];
    }
    // Audited loss, settled after the last launch: a dropped candidate matters only if its key reaches the
    // best threshold ANY list of the row has proven by then.
    if (a.cand_keys && half == 0) a.margin_out[qpos] = margin;
    if (list.lost > kNegInf) atomicMax(a.lost + qpos, seed_enc(list.lost) + SlotList<CAP, NTL>::SLOTS);   // stored keys carry slot bits
  }
}

// overflow[row] |= (a key was dropped for the row) && (no list proved a threshold above it)
__global__ void audit_kernel(const int32_t* seed, const int32_t* lost, uint32_t* overflow, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t l = lost[i];
  if (l > kSeedNone && l >= seed[i]) overflow[i] = 1u;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int pad_dp(int64_t d) {
  if (d <= 128) return 128;
  if (d <= 256) return 256;
  if (d <= 512) return 512;
  if (d <= 1024) return 1024;
  return 0;
}
static int waves_for_dp(int dp) { return dp <= 512 ? 8 : 4; }   // waves that own queries (d <= 1024: four split-k pairs)

int scan_bf16_supported(int64_t d, int kk, int dtype) {
  (void)dtype;
  const int dp = pad_dp(d);
  if (dp == 0) return 0;
  if (kk <= 20) return 1;                          // two 16-entry lists per query keep 14 each after a compaction
  return (kk <= 44 && dp <= 512) ? 1 : 0;          // two 32-entry lists (their 64 KiB do not fit beside 64 KiB tiles: d <= 512)
}

// d <= 1024 keeps the 15-entry lists up to k + self = 20: the split-k pair kernel has no LDS for 16-entry lists beside its
// exchange block, and the one-wave-per-SIMD kernel that had (380-395 VGPRs, 7-17 of them spilled) is retired.  Two 15-entry
// lists hold 13 each after a compaction: enough for the k-th best of their union; the surplus goes to the overflow list.
int scan_bf16_cap(int kk, int dp) {
  if (dp > 512) return B_CAP;
  return kk <= B_CAP - 4 ? B_CAP : (kk <= 20 ? B_CAP_BIG : B_CAP_WIDE);
}
int scan_bf16_slot_ulp(int cap) { return cap <= 16 ? 16 : 32; }
int scan_bf16_dp(int64_t d) { return pad_dp(d); }

int launch_prep_half(const void* X, int64_t n, int64_t d, int dtype, int metric, const float* scal,
                     const uint32_t* max_n, void* Z, int64_t n_pad, int dp, int z_f16, float* zn, float* rn,
                     float* un, float* cb, uint32_t* maxima, hipStream_t s) {
  PrepArgs a{X, n, d, dtype, metric, scal, max_n, Z, n_pad, dp, z_f16, zn, rn, un, cb, maxima};
  int64_t grid = (n_pad + 3) / 4;
  if (grid > 1024) grid = 1024;          // 16 waves per CU; fewer, longer workgroups keep the final atomics few
  hipLaunchKernelGGL(prep_half_kernel, dim3((unsigned)grid), dim3(256), 0, s, a);
  MMF_LAUNCH_CHECK();
  return MMF_OK;
}

static size_t scan_b16_lds(int ks, int nw, int tpb, int cap, bool splitk) {
  const int nqw = splitk ? nw / 2 : nw;
  return (size_t)(2 * tpb) * (B_CT * ks * 32) + (size_t)(2 * tpb) * 64 * 4 + (size_t)cap * (64 * nqw) * 4 +
         (splitk ? (size_t)nqw * 4 * 64 * 16 + 64 : 0);
}

int scan_b16_queries_per_block(int dp) { return 32 * waves_for_dp(dp); }

// splits of a row block that run side by side; the remaining col_splits / conc follow in later rounds
static int conc_splits_for(int64_t row_blocks, int col_splits) {
  int cs = 1;
  while (cs < 8 && cs < col_splits && row_blocks * cs < 256) cs <<= 1;
  return cs;
}
static int64_t scan_b16_round_blocks(int64_t row_blocks, int cs) {
  const int per = 8 / cs;                                 // row blocks per group of 8 block ids
  return ((row_blocks + per - 1) / per) * 8;
}
static int64_t scan_b16_grid(int64_t n_rows, int col_splits, int dp) {
  const int qt = scan_b16_queries_per_block(dp);
  const int64_t row_blocks = (n_rows + qt - 1) / qt;
  const int cs = conc_splits_for(row_blocks, col_splits);
  return scan_b16_round_blocks(row_blocks, cs) * (col_splits / cs);
}

// bytes of the global id-slot scratch ([grid][slots][list threads] u32) a launch needs
size_t scan_b16_scratch_bytes(int64_t n_rows, int col_splits, int dp, int cap) {
  return (size_t)scan_b16_grid(n_rows, col_splits, dp) * (cap <= 16 ? 16 : 32) * (64 * waves_for_dp(dp)) * 4 + 256;
}

template <int KS, int NW, int TPB, int CAP, bool SPLITK = false>
static int launch_b16_t(const ScanB16Args& a, bool f16, int64_t grid, hipStream_t s) {
  const size_t lds = scan_b16_lds(KS, NW, TPB, CAP, SPLITK);
  auto go = [&](auto kern) -> int {
    MMF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * NW), lds, s, a);
    MMF_LAUNCH_CHECK();
    return MMF_OK;
  };
  if (a.debug != 0) {   // instrumented build of the same kernel (MMF_SCAN_DEBUG)
    if (f16) return go(scan_b16x_kernel<KS, true, true, NW, TPB, CAP, SPLITK>);
    return go(scan_b16x_kernel<KS, false, true, NW, TPB, CAP, SPLITK>);
  }
  if (f16) return go(scan_b16x_kernel<KS, true, false, NW, TPB, CAP, SPLITK>);
  return go(scan_b16x_kernel<KS, false, false, NW, TPB, CAP, SPLITK>);
}

// Settles the audited losses of all scan launches of a problem (call once, after the last one).
int launch_scan_b16_audit(const ScanB16Panel& pn, uint32_t* overflow, int64_t n_rows, hipStream_t s) {
  hipLaunchKernelGGL(audit_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, s, pn.seed, pn.seed + pn.seed_stride,
                     overflow, n_rows);
  MMF_LAUNCH_CHECK();
  return MMF_OK;
}

// col_splits must be a power of two.  Lists are indexed by query position, cap = B_CAP.
int launch_scan_b16(const void* ZQ, const void* ZC, const float* cb, const float* q_zn, const float* q_rn,
                    const float* q_un, const uint32_t* maxima, int64_t n_rows, int64_t m, int64_t m_pad, int dp,
                    int64_t d, bool f16, int metric, int kk, int col_splits, const CandLists& L, void* scratch,
                    const ScanB16Panel& pn, hipStream_t s, int* grid_out) {
  ScanB16Args a{};
  a.ZQ = ZQ; a.ZC = ZC; a.cb = cb; a.q_zn = q_zn; a.q_rn = q_rn; a.q_un = q_un; a.maxima = maxima;
  a.n_rows = n_rows; a.m = m; a.tiles_total = m_pad / B_CT;
  a.tiles_per_split = (a.tiles_total + col_splits - 1) / col_splits;
  if (a.tiles_per_split * (int64_t)B_CT * dp * 2 >= (int64_t(1) << 32)) {   // 32-bit scalar offsets of the tile DMA
    set_error("scan_b16: a column range of %lld tiles x %d exceeds 4 GiB of operands; use more col_splits",
              (long long)a.tiles_per_split, dp);
    return MMF_E_UNSUPPORTED;
  }
  a.row_blocks = (n_rows + scan_b16_queries_per_block(dp) - 1) / scan_b16_queries_per_block(dp);
  a.col_splits = col_splits; a.kk = kk; a.metric = metric; a.d = (int)d;
  a.conc_splits = conc_splits_for(a.row_blocks, col_splits);
  a.blocks_per_round = scan_b16_round_blocks(a.row_blocks, a.conc_splits);
  a.lists_total = L.lists; a.list_base = pn.list_base;
  a.seg_len = pn.seg_len; a.seg_stride = pn.seg_stride; a.id_off = pn.id_off;
  a.seed = pn.seed; a.lost = pn.seed + pn.seed_stride; a.share = pn.share;
  if (!pn.seed) { set_error("scan_b16: threshold buffers missing"); return MMF_E_INTERNAL; }
  if (pn.list_base + 2 * col_splits > L.lists) { set_error("scan_b16: list slots out of range"); return MMF_E_INTERNAL; }
  {
    const char* dbg = getenv("MMF_SCAN_DEBUG");
    a.debug = dbg ? atoi(dbg) : 0;
    a.dbg = nullptr;
    if (a.debug & (8 | 16)) {
      static unsigned long long* dbuf = nullptr;
      if (!dbuf) MMF_HIP(hipMalloc(&dbuf, 128));
      MMF_HIP(hipMemsetAsync(dbuf, 0, 128, s));
      a.dbg = dbuf;
    }
  }
  a.cand_cnt = L.cnt; a.cand_ids = L.ids; a.overflow = L.overflow; a.cand_keys = L.keys; a.margin_out = L.margin;
  a.spill_cnt = L.spill_cnt; a.spill_ids = L.spill_ids; a.spill_cap = L.spill_cap;
  a.spill_stacks = L.spill_stacks;
  const int64_t grid = scan_b16_grid(n_rows, col_splits, dp);
  a.lids = reinterpret_cast<uint32_t*>(scratch);
  if (grid_out) *grid_out = (int)grid;
  int rc = MMF_E_INTERNAL;
  const bool big = (L.cap == B_CAP_BIG);   // k + self in 12..20: 16-entry lists, one tile per barrier
  if (L.cap == B_CAP_WIDE) {               // k + self in 21..44: 32-entry lists, one tile per barrier, d <= 512
    switch (dp) {
      case 128: rc = launch_b16_t<8, 8, 1, B_CAP_WIDE>(a, f16, grid, s); break;
      case 256: rc = launch_b16_t<16, 8, 1, B_CAP_WIDE>(a, f16, grid, s); break;
      case 512: rc = launch_b16_t<32, 8, 1, B_CAP_WIDE>(a, f16, grid, s); break;
      default: set_error("scan_b16: 32-entry lists need a padded dim <= 512 (got %d)", dp);
    }
  } else
  switch (dp) {
    // d <= 256: the smaller tiles leave room for eight stages — four tiles per barrier (-1.5 % against two at d = 256)
    case 128: rc = big ? launch_b16_t<8, 8, 2, B_CAP_BIG>(a, f16, grid, s) : launch_b16_t<8, 8, 4, B_CAP>(a, f16, grid, s); break;
    case 256: rc = big ? launch_b16_t<16, 8, 2, B_CAP_BIG>(a, f16, grid, s) : launch_b16_t<16, 8, 4, B_CAP>(a, f16, grid, s); break;
    case 512: rc = big ? launch_b16_t<32, 8, 1, B_CAP_BIG>(a, f16, grid, s) : launch_b16_t<32, 8, 2, B_CAP>(a, f16, grid, s); break;
    case 1024: rc = launch_b16_t<64, 8, 1, B_CAP, true>(a, f16, grid, s); break;
    default: set_error("scan_b16: unsupported padded dim %d", dp);
  }
  if (rc == MMF_OK && (a.debug & 16)) {
    unsigned long long h[16];
    MMF_HIP(hipMemcpyAsync(h, a.dbg, 128, hipMemcpyDeviceToHost, s));
    MMF_HIP(hipStreamSynchronize(s));
    const double tiles = (double)h[3];
    fprintf(stderr, "[mmf scan stamps] per wave-tile cycles: dma-wait %.0f  barrier-wait %.0f  filter %.0f  chain(+DMA issue) %.0f  (sum %.0f)\n",
            h[8] / tiles, h[0] / tiles, h[1] / tiles, h[2] / tiles, (h[8] + h[0] + h[1] + h[2]) / tiles);
    fprintf(stderr, "[mmf scan stamps] list code: cold entries %llu x %.0f cycles, warm entries %llu x %.0f cycles\n",
            h[6], h[6] ? (double)h[4] / h[6] : 0.0, h[7], h[7] ? (double)h[5] / h[7] : 0.0);
  }
  if (rc == MMF_OK && (a.debug & 8)) {
    unsigned long long h[8];
    MMF_HIP(hipMemcpyAsync(h, a.dbg, 64, hipMemcpyDeviceToHost, s));
    MMF_HIP(hipStreamSynchronize(s));
    fprintf(stderr, "[mmf scan dbg] wave-tiles=%llu slow-entries=%llu (cold %llu) hits=%llu warm-compactions=%llu lanes-with-hit=%llu\n",
            h[5], h[0], h[1], h[2], h[3], h[4]);
  }
  return rc;
}

}  // namespace mmf
