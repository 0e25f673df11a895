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
int launch_median_begin_count(void* state, unsigned long long cnt, hipStream_t s) {
  hipLaunchKernelGGL(median_init_kernel, dim3(1), dim3(256), 0, s, reinterpret_cast<MedianState*>(state), (cnt - 1) / 2);
  MMF_LAUNCH_CHECK();
  return MMF_OK;
}
int launch_median_begin(void* state, int64_t n, hipStream_t s) {
  return launch_median_begin_count(state, (unsigned long long)n * (unsigned long long)(n - 1), s);
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

// Lower median (torch.median semantics: element (count - 1) / 2 of the sorted values) of a flat array — the
// similarity statistics of preprocess_hypergraph.py:190-196, 259-265 and the edge-weight median of :885-897.
// The array is walked as rows of 4096 values (plus one ragged row) by the same histogram kernel; a row0 that no
// column index can reach switches the diagonal skip off.
constexpr int64_t kNoDiagonal = -(int64_t(1) << 62);
int launch_lower_median(const float* v, int64_t count, float* out, void* state, hipStream_t s) {
  MMF_TRY(launch_median_begin_count(state, (unsigned long long)count, s));
  const int64_t C = 4096, full = count / C, tail = count - full * C;
  for (int pass = 0; pass < 4; ++pass) {
    if (full > 0) MMF_TRY(launch_median_accumulate(v, C, kNoDiagonal, full, state, pass, s));
    if (tail > 0) MMF_TRY(launch_median_accumulate(v + full * C, tail, kNoDiagonal, 1, state, pass, s));
    MMF_TRY(launch_median_next(state, pass, out, s));
  }
  return MMF_OK;
}

// ------------------------------------------------------------------------------------------------
// mean / std / min / max of a flat f32 array in ONE pass (f64 accumulation around a pivot), deterministic:
// every workgroup writes its partial, one workgroup merges them in index order.
// Replaces: K.mean(), K.std(), K.min(), K.max() — four passes — at preprocess_hypergraph.py:190-195, 260-263.
// ------------------------------------------------------------------------------------------------
struct StatPartial { double s1, s2; float mn, mx; };

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}

__global__ __launch_bounds__(256) void stats_partial_kernel(const float* __restrict__ v, int64_t count, StatPartial* __restrict__ part) {
  const double p = (double)v[0];                       // pivot: sums of (x - p) stay small when the values cluster
  double s1 = 0.0, s2 = 0.0;
  float mn = __builtin_huge_valf(), mx = -__builtin_huge_valf();
  auto feed = [&](float x) {
    const double dx = (double)x - p;
    s1 += dx; s2 = __builtin_fma(dx, dx, s2);
    mn = fminf(mn, x); mx = fmaxf(mx, x);
  };
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  const int64_t n4 = ((reinterpret_cast<uintptr_t>(v) & 15) == 0) ? (count >> 2) : 0;
  for (int64_t i = tid; i < n4; i += nth) {
    const f32x4 x = reinterpret_cast<const f32x4*>(v)[i];
    feed(x[0]); feed(x[1]); feed(x[2]); feed(x[3]);
  }
  for (int64_t i = n4 * 4 + tid; i < count; i += nth) feed(v[i]);
  s1 = wave_sum(s1); s2 = wave_sum(s2);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
  __shared__ StatPartial w[4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) w[wave] = StatPartial{s1, s2, mn, mx};
  __syncthreads();
  if (threadIdx.x == 0) {
    StatPartial r = w[0];
    for (int i = 1; i < 4; ++i) { r.s1 += w[i].s1; r.s2 += w[i].s2; r.mn = fminf(r.mn, w[i].mn); r.mx = fmaxf(r.mx, w[i].mx); }
    part[blockIdx.x] = r;
  }
}

// out[0..3] = mean, std (unbiased, as torch.std), min, max; out[4] is the median's slot.  pivot[0]: the value the
// partial sums were taken around.
__global__ __launch_bounds__(256) void stats_final_kernel(const float* __restrict__ pivot, int64_t count, const StatPartial* __restrict__ part,
                                                          int64_t nparts, double* __restrict__ out) {
  __shared__ StatPartial w[256];
  StatPartial r{0.0, 0.0, __builtin_huge_valf(), -__builtin_huge_valf()};
  for (int64_t i = threadIdx.x; i < nparts; i += 256) {   // fixed assignment, fixed order: bit-reproducible
    r.s1 += part[i].s1; r.s2 += part[i].s2; r.mn = fminf(r.mn, part[i].mn); r.mx = fmaxf(r.mx, part[i].mx);
  }
  w[threadIdx.x] = r;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      w[threadIdx.x].s1 += w[threadIdx.x + o].s1; w[threadIdx.x].s2 += w[threadIdx.x + o].s2;
      w[threadIdx.x].mn = fminf(w[threadIdx.x].mn, w[threadIdx.x + o].mn); w[threadIdx.x].mx = fmaxf(w[threadIdx.x].mx, w[threadIdx.x + o].mx);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double n = (double)count, p = (double)pivot[0];
    const double mean = p + w[0].s1 / n;
    double var = (w[0].s2 - w[0].s1 * w[0].s1 / n) / (n - 1.0);   // count == 1: 0 / 0 = NaN, as torch.std
    if (var < 0.0) var = 0.0;
    out[0] = mean; out[1] = __builtin_sqrt(var); out[2] = (double)w[0].mn; out[3] = (double)w[0].mx;
  }
}

__global__ void stats_median_kernel(const float* med, double* out) { out[4] = (double)*med; }

size_t array_stats_scratch_bytes() { return 2048 * sizeof(StatPartial) + sizeof(MedianState) + 256; }

int launch_array_stats(const float* v, int64_t count, double* out, void* scratch, hipStream_t s) {
  StatPartial* part = reinterpret_cast<StatPartial*>(scratch);
  char* rest = reinterpret_cast<char*>(scratch) + 2048 * sizeof(StatPartial);
  float* med = reinterpret_cast<float*>(rest);
  void* mstate = rest + 256;
  int64_t grid = (count + 256 * 16 - 1) / (256 * 16);
  if (grid > 2048) grid = 2048;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(stats_partial_kernel, dim3((unsigned)grid), dim3(256), 0, s, v, count, part);
  MMF_LAUNCH_CHECK();
  MMF_TRY(launch_stats_finish(part, grid, v, count, out, s));
  MMF_TRY(launch_lower_median(v, count, med, mstate, s));
  return launch_stats_set_median(med, out, s);
}

// the merge of per-workgroup partials (StatPartial: {double s1, s2; float mn, mx}) written by ANY producer kernel
int launch_stats_finish(const void* part, int64_t nparts, const float* pivot, int64_t count, double* out, hipStream_t s) {
  hipLaunchKernelGGL(stats_final_kernel, dim3(1), dim3(256), 0, s, pivot, count, reinterpret_cast<const StatPartial*>(part), nparts, out);
  MMF_LAUNCH_CHECK();
  return MMF_OK;
}
int launch_stats_set_median(const float* med, double* out, hipStream_t s) {
  hipLaunchKernelGGL(stats_median_kernel, dim3(1), dim3(1), 0, s, med, out);
  MMF_LAUNCH_CHECK();
  return MMF_OK;
}
size_t stat_partial_bytes() { return sizeof(StatPartial); }
// one radix pass over `rows` rows of `cols` values each (a panel of a rectangular matrix: no diagonal to skip)
int launch_median_accumulate_flat(const float* K, int64_t cols, int64_t rows, void* state, int pass, hipStream_t s) {
  return launch_median_accumulate(K, cols, kNoDiagonal, rows, state, pass, s);
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
  const float* kr = K + row * n;
  uint32_t c = 0;
  if (((n & 3) == 0) && ((reinterpret_cast<uintptr_t>(K) & 15) == 0)) {       // 16 bytes per lane, 1 KiB per wave instruction
    for (int64_t j = (int64_t)lane * 4; j < n; j += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(kr + j);
      c += (v[0] < thr ? 0u : 1u) + (v[1] < thr ? 0u : 1u) + (v[2] < thr ? 0u : 1u) + (v[3] < thr ? 0u : 1u);   // skipped iff K < thr (:198)
    }
  } else {
    for (int64_t j = lane; j < n; j += 64) c += (kr[j] < thr) ? 0u : 1u;
  }
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
  const float* kr = K + row * n;
  unsigned long long pos = row_off[row];
  // Four 64-column groups per trip: the four loads are in flight together, and every store instruction writes the
  // kept entries of one group to CONSECUTIVE positions (a lane-owns-4-columns layout would make the stores strided).
  int64_t j0 = 0;
  for (; j0 + 256 <= n; j0 += 256) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = kr[j0 + 64 * u + lane];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool keep = !(v[u] < thr);
      const unsigned long long mask = __ballot(keep);
      if (keep) {
        const unsigned long long p = pos + (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull));
        if ((int64_t)p < capacity) { ei_row[p] = row0 + row; ei_col[p] = j0 + 64 * u + lane; ew[p] = v[u]; }
      }
      pos += (unsigned long long)__popcll(mask);
    }
  }
  for (; j0 < n; j0 += 64) {
    const int64_t j = j0 + lane;
    const float v = (j < n) ? kr[j] : 0.0f;
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
