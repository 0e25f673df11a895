// mmf_order.hip — the ORDER in which the 16-bit scan takes its query rows.
//
// A wave of the scan (mmf_scan_bf16.hip) owns 32 consecutive query rows and leaves its matrix-core loop for the list code
// whenever ANY of them has a column inside its margin in the current tile.  On near-duplicate data (patch embeddings: tight
// clusters) a row's margin band is its whole cluster.  With the rows in the caller's order the 32 queries of a wave belong to 32
// different clusters and their hits fall into 32 x |cluster| different tiles (42 % of all wave-tiles at the benchmark's clustered
// workload); with near-duplicate rows NEXT to each other the hits of a wave coincide and the same work takes a single visit.
// Which wave scans for a row changes nothing in the row's result, so the scan may take its queries in any order:
//
//   order_keys   every query's nearest and second-nearest of 128 pivot rows (every (n / 128)-th query row), by the cosine of the
//                16-bit operands on the matrix cores: key = pivot << 25 | second pivot << 18 | cosine in 18 bits.
//                Near-duplicate rows agree on both pivots and, to ~1e-3, on the cosine — whether or not a pivot is one of
//                them.  (Measured on the benchmark's clustered workload, scripts/query_order_purity.py: 1.5 clusters per
//                32-row wave; 31.8 in the generator's order, 1.24 at best; one pivot id alone needs 1024 pivots for that.)
//   radix sort   of (key, row)  ->  perm: scan position -> row
//   gather       the query-side operands (16-bit rows, three norms) into that order
//
// The re-rank maps a scan position back to its row (SelectProblem::perm), outputs stay in the caller's row order, bits unchanged.
// It pays only where rows have near-duplicates: order_keys on a sample of the rows counts those within cosine 0.965 of a pivot
// other than themselves, and the host applies the order when the estimate for all rows exceeds eight per pivot (Gaussian rows: none).
// No reference counterpart: the reference materialises the N x N matrix (similarity_kernel.py:37-107) and has no scan.
#include <hipcub/hipcub.hpp>

#include "mmf_dev.h"
#include "mmf_host.h"

namespace mmf {

constexpr int QO_P = 128;          // pivots
constexpr int QO_W = 8;            // waves per workgroup, 32 queries each
constexpr int QO_KC = 64;          // k per staged chunk of the pivot rows
// cosine from which a row counts as a near-duplicate of a pivot row.  The order pays when a row's margin band is (most of) its cluster; on
// looser clusters it COSTS (scripts/query_order_loose.py, N = 262144 in 2048 clusters, scan + re-rank with the order off -> on, by the
// cosine inside a cluster: 0.999: 73 -> 57 ms; 0.99: 74 -> 62; 0.978: 69 -> 62; 0.969: 63.6 -> 59.9; 0.962: 58.0 -> 58.5; 0.92: 56.2 -> 58.3;
// 0.8: 55.5 -> 57.2), so the probe asks for near-DUPLICATES, not for neighbours.
constexpr float QO_NEAR = 0.965f;
constexpr int QO_LD = QO_KC + 8;   // 16-bit elements per LDS row (144 B: ds_read_b128 of 32 consecutive rows spreads over the banks)

typedef __bf16 qo_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 qo_f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t qo_u32x4 __attribute__((ext_vector_type(4)));

struct OrderArgs {
  const uint16_t* ZQ; const float* q_zn; int64_t n;
  int dp;
  uint32_t* key; uint32_t* row; uint32_t* near_cnt;
  int64_t wg_stride;       // workgroup b takes the 256 queries from 256 * b * wg_stride on (> 1: a sample, for the count alone)
};

__device__ __forceinline__ int64_t pivot_row(int j, int64_t m) { return ((int64_t)j * m) / QO_P; }

template <bool F16>
__global__ __launch_bounds__(64 * QO_W) void order_keys_kernel(OrderArgs a) {
  __shared__ __attribute__((aligned(16))) uint16_t ptile[QO_P][QO_LD];
  __shared__ float pinv[QO_P];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t q0 = ((int64_t)blockIdx.x * a.wg_stride * QO_W + w) * 32;
  if (tid < QO_P) {
    const float z = a.q_zn[pivot_row(tid, a.n)];
    pinv[tid] = z > 0.0f ? 1.0f / z : 0.0f;
  }
  f32x16 acc[QO_P / 32];
#pragma unroll
  for (int t = 0; t < QO_P / 32; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  // operand rows of this lane: query q0 + (lane & 31), k offset 8 (lane >> 5) inside a 16-wide step
  const uint16_t* qrow = a.ZQ + (q0 + (lane & 31)) * a.dp + 8 * (lane >> 5);
  for (int k0 = 0; k0 < a.dp; k0 += QO_KC) {
    constexpr int PU = QO_P * (QO_KC / 8) / (64 * QO_W);           // pivot rows x 8 units of 16 bytes, per thread
    qo_u32x4 pv[PU], qv[4];
#pragma unroll
    for (int i = 0; i < PU; ++i) {
      const int u = tid + 64 * QO_W * i;
      pv[i] = *reinterpret_cast<const qo_u32x4*>(a.ZQ + pivot_row(u >> 3, a.n) * a.dp + k0 + 8 * (u & 7));
    }
#pragma unroll
    for (int sx = 0; sx < 4; ++sx) qv[sx] = *reinterpret_cast<const qo_u32x4*>(qrow + k0 + 16 * sx);
    __syncthreads();                                                // the previous chunk has been read
#pragma unroll
    for (int i = 0; i < PU; ++i) {
      const int u = tid + 64 * QO_W * i;
      *reinterpret_cast<qo_u32x4*>(&ptile[u >> 3][8 * (u & 7)]) = pv[i];
    }
    __syncthreads();
#pragma unroll
    for (int sx = 0; sx < 4; ++sx) {
#pragma unroll
      for (int t = 0; t < QO_P / 32; ++t) {
        const qo_u32x4 pa = *reinterpret_cast<const qo_u32x4*>(&ptile[32 * t + (lane & 31)][16 * sx + 8 * (lane >> 5)]);
        if constexpr (F16)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(qo_f16x8, pa), __builtin_bit_cast(qo_f16x8, qv[sx]), acc[t], 0, 0, 0);
        else
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(qo_bf16x8, pa), __builtin_bit_cast(qo_bf16x8, qv[sx]), acc[t], 0, 0, 0);
      }
    }
  }
  // C layout: column (query) = lane & 31, row (pivot of the tile) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  // nearest and second-nearest pivot (ties: the smaller pivot)
  float best = -3.0e38f, sec = -3.0e38f;
  int bp = 0, sp = 0;
#pragma unroll
  for (int t = 0; t < QO_P / 32; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const float v = acc[t][r] * pinv[p];
      if (v > best) { sec = best; sp = bp; best = v; bp = p; }
      else if (v > sec) { sec = v; sp = p; }
    }
  {
    const float ob = __shfl_xor(best, 32), os = __shfl_xor(sec, 32);
    const int op = __shfl_xor(bp, 32), osp = __shfl_xor(sp, 32);
    const bool mine = best > ob || (best == ob && bp < op);          // this half's best wins
    const float lb = mine ? ob : best; const int lp = mine ? op : bp;        // the losing best
    const float ws = mine ? sec : os; const int wp = mine ? sp : osp;         // the winner's own second
    if (!mine) { best = ob; bp = op; }
    if (lb > ws || (lb == ws && lp < wp)) { sec = lb; sp = lp; } else { sec = ws; sp = wp; }
  }
  const int64_t q = q0 + (lane & 31);
  bool near = false;
  if (lane < 32 && q < a.n) {
    const float zq = a.q_zn[q];
    float c = zq > 0.0f ? best / zq : 0.0f;
    c = c != c ? 0.0f : (c < -1.0f ? -1.0f : (c > 1.0f ? 1.0f : c));
    // a pivot row is its own nearest pivot: what counts as its near-duplicate is the second one
    const float cn = (q == pivot_row(bp, a.n)) ? (zq > 0.0f ? sec / zq : 0.0f) : c;
    near = cn >= QO_NEAR;
    if (a.key) {
      a.key[q] = ((uint32_t)bp << 25) | ((uint32_t)sp << 18) | (uint32_t)((c + 1.0f) * 131071.5f);
      a.row[q] = (uint32_t)q;
    }
  }
  const int nn = __popcll(__ballot(near));
  if (lane == 0 && nn && a.near_cnt) atomicAdd(a.near_cnt + (blockIdx.x & 63), (uint32_t)nn);
}

// Query-side operands in scan order: position p takes row perm[p]; positions from n up to n_pad are zero rows.
// (Tried: dealing the sorted order out wave by wave, so that the eight waves of a workgroup come from eight distant stretches of it —
// a wave has to be homogeneous, a workgroup need not be, and workgroups of like queries drift apart along the column stream: the
// clustered bench workload shows 109 GB of L2 misses per scan launch instead of 9, `profiles/r03c_pmc_summary.json`.  It is slower,
// 53.4 -> 54.4 ms: eight like waves are in the list code AT THE SAME tiles, so they wait for each other once, not eight times.)
struct GatherArgs {
  const uint16_t* Z; const float* zn; const float* rn; const float* un;
  const uint32_t* perm; int64_t n, n_pad; int dp;
  uint16_t* Zo; float* zno; float* rno; float* uno;
};

__global__ __launch_bounds__(256) void order_gather_kernel(GatherArgs a) {
  const int upr = a.dp >> 3;                                       // 16-byte units per row
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= a.n_pad * upr) return;
  const int64_t pos = g / upr;
  const int part = (int)(g - pos * upr);
  qo_u32x4 v = {0u, 0u, 0u, 0u};
  if (pos < a.n) {
    const int64_t src = (int64_t)a.perm[pos];
    v = *reinterpret_cast<const qo_u32x4*>(a.Z + src * a.dp + 8 * part);
    if (part == 0) { a.zno[pos] = a.zn[src]; a.rno[pos] = a.rn[src]; a.uno[pos] = a.un[src]; }
  } else if (part == 0) {
    a.zno[pos] = 0.0f; a.rno[pos] = 0.0f; a.uno[pos] = 0.0f;
  }
  *reinterpret_cast<qo_u32x4*>(a.Zo + pos * a.dp + 8 * part) = v;
}

static const int32_t* g_last_perm = nullptr;
static int64_t g_last_perm_n = 0;

static size_t order_sort_temp(int64_t n) {
  size_t tb = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                           (uint32_t*)nullptr, (int)n, 0, 32, (hipStream_t)0);
  return (tb + 255) & ~size_t(255);
}

// scratch of query_order_*: four u32 arrays of n (rounded to 64), 64 counters, the sort's temporary storage
size_t query_order_bytes(int64_t n) {
  const size_t nn = ((size_t)n + 63) & ~size_t(63);
  return (4 * nn + 64) * 4 + order_sort_temp(n) + 256;
}

static int launch_order_keys(bool f16, int64_t grid, hipStream_t s, const OrderArgs& a) {
  if (f16) hipLaunchKernelGGL(order_keys_kernel<true>, dim3((unsigned)grid), dim3(64 * QO_W), 0, s, a);
  else hipLaunchKernelGGL(order_keys_kernel<false>, dim3((unsigned)grid), dim3(64 * QO_W), 0, s, a);
  MMF_LAUNCH_CHECK();
  return MMF_OK;
}

// Do the query rows have near-duplicates among themselves?  A sample of up to 32 blocks of 256 rows, spread over the rows, against the
// pivot rows: *near = estimated number of rows within cosine 0.965 of a pivot row other than themselves.  Synchronises the
// stream (the caller decides on the host whether the order is worth its 0.4 ms).
int launch_query_order_probe(const uint16_t* ZQ, const float* q_zn, int64_t n, int dp, bool f16, void* scratch, int64_t* near, hipStream_t s) {
  if (n <= 0 || dp % QO_KC != 0) { set_error("query order: bad shape (n = %lld, dp = %d)", (long long)n, dp); return MMF_E_INTERNAL; }
  const size_t nn = ((size_t)n + 63) & ~size_t(63);
  uint32_t* o = static_cast<uint32_t*>(scratch);
  const int64_t wgs = (n + 32 * QO_W - 1) / (32 * QO_W);
  const int64_t stride = wgs > 32 ? wgs / 32 : 1;
  const int64_t grid = (wgs + stride - 1) / stride;
  OrderArgs a{ZQ, q_zn, n, dp, nullptr, nullptr, o + 4 * nn, stride};
  MMF_HIP(hipMemsetAsync(a.near_cnt, 0, 256, s));
  MMF_TRY(launch_order_keys(f16, grid, s, a));
  uint32_t h[64];
  MMF_HIP(hipMemcpyAsync(h, a.near_cnt, 256, hipMemcpyDeviceToHost, s));
  MMF_HIP(hipStreamSynchronize(s));
  int64_t tot = 0, rows = 0;
  for (uint32_t v : h) tot += v;
  for (int64_t b = 0; b < grid; ++b) { const int64_t r0 = b * stride * 32 * QO_W; rows += (n - r0 < 32 * QO_W) ? (n - r0) : 32 * QO_W; }
  *near = rows > 0 ? (int64_t)((double)tot * (double)n / (double)rows + 0.5) : 0;
  return MMF_OK;
}

int query_order_pivots() { return QO_P; }

// Keys of all rows, their sort, and the query side gathered into that order.  *perm (device, n entries, inside `scratch`): scan
// position -> row.
int launch_query_order_apply(const uint16_t* ZQ, const float* q_zn, const float* q_rn, const float* q_un, int64_t n, int64_t n_pad,
                             int dp, bool f16, void* scratch, uint16_t* Zo, float* zno, float* rno, float* uno, const int32_t** perm,
                             hipStream_t s) {
  const size_t nn = ((size_t)n + 63) & ~size_t(63);
  uint32_t* o = static_cast<uint32_t*>(scratch);
  OrderArgs a{ZQ, q_zn, n, dp, o, o + nn, nullptr, 1};
  MMF_TRY(launch_order_keys(f16, (n + 32 * QO_W - 1) / (32 * QO_W), s, a));
  size_t tb = order_sort_temp(n);
  MMF_HIP(hipcub::DeviceRadixSort::SortPairs(o + 4 * nn + 64, tb, o, o + 2 * nn, o + nn, o + 3 * nn, (int)n, 0, 32, s));
  GatherArgs g{ZQ, q_zn, q_rn, q_un, o + 3 * nn, n, n_pad, dp, Zo, zno, rno, uno};
  const int64_t units = n_pad * (dp >> 3);
  hipLaunchKernelGGL(order_gather_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, s, g);
  MMF_LAUNCH_CHECK();
  *perm = reinterpret_cast<const int32_t*>(o + 3 * nn);
  g_last_perm = *perm; g_last_perm_n = n;
  return MMF_OK;
}

void query_order_forget() { g_last_perm = nullptr; g_last_perm_n = 0; }

// diagnostics (scripts/query_order_purity.py): the permutation of the most recent ordered call, while its workspace is alive
int query_order_last(int32_t* out_host, int64_t n) {
  if (!g_last_perm || n != g_last_perm_n) { set_error("query order: no permutation of %lld rows on record", (long long)n); return MMF_E_INVALID; }
  MMF_HIP(hipMemcpy(out_host, g_last_perm, (size_t)n * 4, hipMemcpyDeviceToHost));
  return MMF_OK;
}

}  // namespace mmf
