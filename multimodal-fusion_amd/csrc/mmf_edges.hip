// mmf_edges.hip — threshold edge builder over a dense [n,n] similarity (row a4 of SURVEY.md §8).
//
//  * lower median of the n(n-1) off-diagonal entries: 4-pass 8-bit radix select on the
//    order-preserving integer image of the floats; HBM-bound (4 reads of K), no sort, no copy.
//    Replaces: K[~eye] gather + torch.median, build_hypergraph/similarity_kernel.py:183-186.
//  * row-major stream compaction of the entries that are NOT below the threshold (self loops
//    kept): count per row -> exclusive scan -> ordered fill with wave ballots.
//    Replaces: the Python double loop with .item(), build_hypergraph/similarity_kernel.py:193-202.
#include "mmf_dev.h"
#include "mmf_host.h"

namespace mmf {

__device__ __forceinline__ uint32_t f2ord(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
  const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}

// state (u64 units): hist[256], then prefix (low 32 bits), then rank
struct MedianState {
  unsigned long long hist[256];
  unsigned long long prefix;
  unsigned long long rank;
};

__global__ void median_init_kernel(MedianState* st, unsigned long long rank) {
  if (threadIdx.x < 256) st->hist[threadIdx.x] = 0ull;
  if (threadIdx.x == 0) { st->prefix = 0ull; st->rank = rank; }
}

// One radix pass: histogram of byte (shift) over the entries whose higher bytes equal the prefix found so far.
// A workgroup walks whole rows (no per-element division), 16 bytes per lane when n % 4 == 0.  Similarities are
// concentrated — in the first passes nearly every entry lands in one or two bins — so a lane counts runs of equal
// bins in a register and touches the LDS histogram only when the bin changes.
// K: rows [row0, row0 + rows) of the n x n matrix (row0 = 0, rows = n: all of it).
__global__ __launch_bounds__(256) void median_hist_kernel(const float* __restrict__ K, int64_t n, int64_t row0, int64_t rows,
                                                          MedianState* st, int shift) {
  __shared__ unsigned int lh[256];
  lh[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t prefix = (uint32_t)st->prefix;
  const uint32_t himask = (shift == 24) ? 0u : (0xffffffffu << (shift + 8));
  const uint32_t want = prefix & himask;
  uint32_t cur = 0xffffffffu, run = 0;
  auto feed = [&](float v) {
    const uint32_t o = f2ord(v);
    if ((o & himask) != want) return;
    const uint32_t bin = (o >> shift) & 255u;
    if (bin == cur) { ++run; return; }
    if (run) atomicAdd(&lh[cur], run);
    cur = bin; run = 1;
  };
  const bool vec = ((n & 3) == 0) && ((reinterpret_cast<uintptr_t>(K) & 15) == 0);
  for (int64_t li = blockIdx.x; li < rows; li += gridDim.x) {
    const float* row = K + li * n;
    const int64_t i = row0 + li;                       // global row: its diagonal entry is skipped
    if (vec) {
      for (int64_t j = (int64_t)threadIdx.x * 4; j < n; j += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + j);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (j + e != i) feed(v[e]);
      }
    } else {
      for (int64_t j = threadIdx.x; j < n; j += 256)
        if (j != i) feed(row[j]);
    }
  }
  if (run) atomicAdd(&lh[cur], run);
  __syncthreads();
  if (lh[threadIdx.x]) atomicAdd(&st->hist[threadIdx.x], (unsigned long long)lh[threadIdx.x]);
}

__global__ void median_pick_kernel(MedianState* st, int shift, float* out) {
  if (threadIdx.x == 0) {
    unsigned long long r = st->rank;
    int b = 0;
    for (; b < 255; ++b) {
      const unsigned long long c = st->hist[b];
      if (r < c) break;
      r -= c;
    }
    st->rank = r;
    st->prefix = st->prefix | ((unsigned long long)b << shift);
    if (shift == 0) *out = ord2f((uint32_t)st->prefix);
  }
  __syncthreads();
  if (threadIdx.x < 256) st->hist[threadIdx.x] = 0ull;
}

int launch_offdiag_lower_median(const float* K, int64_t n, float* out, uint32_t* scratch, hipStream_t s) {
  MedianState* st = reinterpret_cast<MedianState*>(scratch);
  MMF_TRY(launch_median_begin(st, n, s));
  for (int pass = 0; pass < 4; ++pass) {
    MMF_TRY(launch_median_accumulate(K, n, 0, n, st, pass, s));
    MMF_TRY(launch_median_next(st, pass, out, s));
  }
  return MMF_OK;
}

// The same radix select in pieces, for matrices that are recomputed panel by panel instead of stored:
// begin; then for pass 0..3 { accumulate every panel; next }.  `out` is written by next(pass 3).
size_t median_state_bytes() { return sizeof(MedianState); }
int launch_median_begin(void* state, int64_t n, hipStream_t s) {
  const unsigned long long cnt = (unsigned long long)n * (unsigned long long)(n - 1);
  hipLaunchKernelGGL(median_init_kernel, dim3(1), dim3(256), 0, s, reinterpret_cast<MedianState*>(state), (cnt - 1) / 2);
  MMF_LAUNCH_CHECK();
  return MMF_OK;
}
int launch_median_accumulate(const float* K, int64_t n, int64_t row0, int64_t rows, void* state, int pass, hipStream_t s) {
  if (rows <= 0) return MMF_OK;
  int64_t grid = rows < 4096 ? rows : 4096;          // workgroups take whole rows
  hipLaunchKernelGGL(median_hist_kernel, dim3((unsigned)grid), dim3(256), 0, s, K, n, row0, rows,
                     reinterpret_cast<MedianState*>(state), 24 - 8 * pass);
  MMF_LAUNCH_CHECK();
  return MMF_OK;
}
int launch_median_next(void* state, int pass, float* out, hipStream_t s) {
  hipLaunchKernelGGL(median_pick_kernel, dim3(1), dim3(256), 0, s, reinterpret_cast<MedianState*>(state), 24 - 8 * pass, out);
  MMF_LAUNCH_CHECK();
  return MMF_OK;
}

// ------------------------------------------------------------------------------------------------
// threshold edges
// ------------------------------------------------------------------------------------------------
// K: `rows` rows of n entries each (a panel of the square matrix, or all of it)
__global__ __launch_bounds__(256) void thr_count_kernel(const float* __restrict__ K, int64_t n, int64_t rows, float thr,
                                                        uint32_t* __restrict__ row_cnt) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  uint32_t c = 0;
  for (int64_t j = lane; j < n; j += 64) c += (K[row * n + j] < thr) ? 0u : 1u;  // skipped iff K < thr (:198)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor((int)c, o);
  if (lane == 0) row_cnt[row] = c;
}

// Exclusive offsets of the rows, starting at the edges already counted (*out_count on entry: 0 for a whole matrix,
// the running total when panels are processed one after the other); *out_count is advanced by this panel's edges.
__global__ __launch_bounds__(1024) void thr_scan_kernel(const uint32_t* __restrict__ row_cnt, int64_t n,
                                                        unsigned long long* __restrict__ row_off,
                                                        int64_t* __restrict__ out_count) {
  __shared__ unsigned long long part[1024];
  const unsigned long long base = (unsigned long long)*out_count;
  const int t = threadIdx.x;
  const int64_t per = (n + 1023) / 1024;
  const int64_t b = (int64_t)t * per;
  int64_t e = b + per;
  if (e > n) e = n;
  unsigned long long sum = 0;
  for (int64_t i = b; i < e; ++i) sum += row_cnt[i];
  part[t] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    unsigned long long v = (t >= o) ? part[t - o] : 0ull;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  unsigned long long run = base + ((t == 0) ? 0ull : part[t - 1]);
  for (int64_t i = b; i < e; ++i) { row_off[i] = run; run += row_cnt[i]; }
  __syncthreads();                                   // everybody has read the base
  if (t == 1023) *out_count = (int64_t)(base + part[1023]);
}

__global__ __launch_bounds__(256) void thr_fill_kernel(const float* __restrict__ K, int64_t n, int64_t row0, int64_t rows,
                                                       float thr, const unsigned long long* __restrict__ row_off,
                                                       int64_t* __restrict__ ei_row, int64_t* __restrict__ ei_col,
                                                       float* __restrict__ ew, int64_t capacity) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  unsigned long long pos = row_off[row];
  for (int64_t j0 = 0; j0 < n; j0 += 64) {
    const int64_t j = j0 + lane;
    const float v = (j < n) ? K[row * n + j] : 0.0f;
    const bool keep = (j < n) && !(v < thr);
    const unsigned long long mask = __ballot(keep);
    if (keep) {
      const unsigned long long p = pos + (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull));
      if ((int64_t)p < capacity) { ei_row[p] = row0 + row; ei_col[p] = j; ew[p] = v; }
    }
    pos += (unsigned long long)__popcll(mask);
  }
}

// K: rows [row0, row0 + rows) of the n x n matrix.  Edges are written to (ei_row, ei_col, ew)[0 .. count), row-major.
int launch_threshold_edges_panel(const float* K, int64_t n, int64_t row0, int64_t rows, float thr, int64_t* ei_row,
                                 int64_t* ei_col, float* ew, int64_t capacity, int64_t* out_count, uint32_t* scratch,
                                 size_t scratch_u32, hipStream_t s) {
  if ((size_t)(2 * (rows + 1) + rows) > scratch_u32) { set_error("threshold_edges: scratch too small"); return MMF_E_INTERNAL; }
  unsigned long long* row_off = reinterpret_cast<unsigned long long*>(scratch);
  uint32_t* row_cnt = scratch + 2 * (rows + 1);
  const unsigned grid = (unsigned)((rows + 3) / 4);
  hipLaunchKernelGGL(thr_count_kernel, dim3(grid), dim3(256), 0, s, K, n, rows, thr, row_cnt);
  MMF_LAUNCH_CHECK();
  hipLaunchKernelGGL(thr_scan_kernel, dim3(1), dim3(1024), 0, s, row_cnt, rows, row_off, out_count);
  MMF_LAUNCH_CHECK();
  if (capacity > 0) {
    hipLaunchKernelGGL(thr_fill_kernel, dim3(grid), dim3(256), 0, s, K, n, row0, rows, thr, row_off, ei_row, ei_col, ew, capacity);
    MMF_LAUNCH_CHECK();
  }
  return MMF_OK;
}

int launch_threshold_edges(const float* K, int64_t n, float thr, int64_t* ei, float* ew, int64_t capacity,
                           int64_t* out_count, uint32_t* scratch, size_t scratch_u32, hipStream_t s) {
  return launch_threshold_edges_panel(K, n, 0, n, thr, ei, ei ? ei + capacity : nullptr, ew, capacity, out_count, scratch,
                                      scratch_u32, s);
}

}  // namespace mmf
