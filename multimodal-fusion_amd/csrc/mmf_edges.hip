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

// a first-row index no column can reach: switches the diagonal skip off (flat arrays, rectangular panels)
constexpr int64_t kNoDiagonal = -(int64_t(1) << 62);
int64_t median_no_diagonal_row() { return kNoDiagonal; }

// The histogram of a pass exists in 16 copies (a workgroup adds to copy blockIdx % 16): similarities are concentrated, so
// every workgroup ends with the same one or two non-empty bins, and thousands of atomics on ONE address queue up at the
// L2 (30 us per pass over 36 MB, measured); the pick sums the copies.
constexpr int kHistCopies = 16;
struct MedianState {
  unsigned long long hist[kHistCopies][256];
  unsigned long long prefix;
  unsigned long long rank;
};

__global__ __launch_bounds__(256) void median_init_kernel(MedianState* st, unsigned long long rank) {
  for (int c = 0; c < kHistCopies; ++c) st->hist[c][threadIdx.x] = 0ull;
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
  if (lh[threadIdx.x]) atomicAdd(&st->hist[blockIdx.x % kHistCopies][threadIdx.x], (unsigned long long)lh[threadIdx.x]);
}

// the bin that holds the wanted rank: 256 threads, inclusive scan of the histogram in LDS
__global__ __launch_bounds__(256) void median_pick_kernel(MedianState* st, int shift, float* out) {
  __shared__ unsigned long long cum[256];
  const int t = threadIdx.x;
  unsigned long long mine = 0ull;
  for (int c = 0; c < kHistCopies; ++c) { mine += st->hist[c][t]; st->hist[c][t] = 0ull; }
  cum[t] = mine;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const unsigned long long v = (t >= o) ? cum[t - o] : 0ull;
    __syncthreads();
    cum[t] += v;
    __syncthreads();
  }
  const unsigned long long r = st->rank;
  const unsigned long long before = cum[t] - mine;
  // the first bin whose inclusive count exceeds the rank (bin 255 if none does: the serial loop's behaviour)
  const bool here = (r >= before && r < cum[t]) || (t == 255 && r >= cum[255]);
  __syncthreads();
  if (here) {
    st->rank = r - before;
    st->prefix = st->prefix | ((unsigned long long)t << shift);
    if (shift == 0) *out = ord2f((uint32_t)st->prefix);
  }
}

int launch_offdiag_lower_median(const float* K, int64_t n, float* out, void* scratch /* median_scratch_bytes(n (n - 1)) */,
                                hipStream_t s) {
  const unsigned long long count = (unsigned long long)n * (unsigned long long)(n - 1);
  return lower_median_of(
      count, [&](float* sample, int sc) { return launch_sample_gather(K, n, count, sample, sc, s); },
      [&](const MedianConsume& consume) { return consume(K, n, 0, n); }, out, scratch, s);
}

// The same radix select in pieces, for matrices that are recomputed panel by panel instead of stored:
// begin; then for pass 0..3 { accumulate every panel; next }.  `out` is written by next(pass 3).
int launch_median_begin_count(void* state, unsigned long long cnt, hipStream_t s) {
  hipLaunchKernelGGL(median_init_kernel, dim3(1), dim3(256), 0, s, reinterpret_cast<MedianState*>(state), (cnt - 1) / 2);
  MMF_LAUNCH_CHECK();
  return MMF_OK;
}
int launch_median_accumulate(const float* K, int64_t n, int64_t row0, int64_t rows, void* state, int pass, hipStream_t s) {
  if (rows <= 0) return MMF_OK;
  int64_t grid = rows < 2048 ? rows : 2048;          // workgroups take whole rows (fewer workgroups: fewer same-address atomics at the end)
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
// Lower median in ONE sweep over the data (instead of four): a sampled bracket, verified by exact counts.
//
//  1. 32768 entries at hashed positions (uniform over the population, so its quantiles are unbiased whatever the
//     row / column structure of the matrix) go through an in-LDS radix select that returns the sample's quantiles at
//     0.5 -+ 1.66 % (six standard deviations of a sample quantile): the bracket [lo, hi].
//  2. The sweep counts the entries below lo and appends the entries inside [lo, hi] (about 3.3 % of them) to a buffer:
//     a wave compacts its hits through a private LDS stage (ballot ranks, no atomics) and reserves buffer space with one
//     atomicAdd per ~800 entries.
//  3. If  below <= rank < below + inside  (and the buffer held everything), the median is the entry of rank
//     rank - below among the buffered values: a radix select over 3 % of the data.  The result is EXACT — the sample
//     only chose where to look.  Otherwise (an unlucky bracket, or so many equal values that the buffer overflows) the
//     caller falls back to the four-pass radix select; both give the same value.
// The sweep reads the data once, so the matrices that are recomputed panel by panel (mmf_combined_offdiag_median,
// mmf_sim_dense_stats without an output) are recomputed once instead of four times.
// ------------------------------------------------------------------------------------------------
constexpr int kBracketSample = 32768;
constexpr int kBracketHalfWidth = 544;          // ranks either side of the sample's middle: 3 sqrt(32768)

constexpr int kBracketSegs = 64;                 // the buffer is 64 segments with a cursor each: waves that reserve space do not
constexpr int kBracketPad = 16;                  //   queue up on ONE address (cursors and below-counters sit 128 B apart)
struct BracketState {
  unsigned long long rank, seg_cap, inside, pad0;
  float lo, hi;
  uint32_t fail, pad;
  MedianState sel;                               // radix select inside the buffer
  unsigned long long cursor[kBracketSegs * kBracketPad];
  unsigned long long below[kBracketSegs * kBracketPad];
};

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {      // splitmix64 finaliser
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// sample entry i of a stored array: flat (n_sq == 0: `count` values) or the off-diagonal of an n_sq x n_sq matrix
__global__ __launch_bounds__(256) void sample_gather_kernel(const float* __restrict__ data, int64_t n_sq, unsigned long long count,
                                                            int s, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= s) return;
  const unsigned long long idx = mix64((unsigned long long)i) % count;
  if (n_sq == 0) { out[i] = data[idx]; return; }
  const unsigned long long r = idx / (unsigned long long)(n_sq - 1);
  unsigned long long c = idx - r * (unsigned long long)(n_sq - 1);
  c += (c >= r) ? 1ull : 0ull;
  out[i] = data[r * (unsigned long long)n_sq + c];
}

// sample entry i of a matrix that is NOT stored: exp(-lambda |a_r - b_c|^2) [* exp(-lambda_g |p_r - p_c|^2)] of a hashed
// pair, one wave per pair, lanes across k (not the canonical chain: the sample only places the bracket).
// offdiag != 0: A == B, pairs (r, c) with r != c.
__global__ __launch_bounds__(256) void sample_pairs_kernel(const void* __restrict__ A, const void* __restrict__ B, int64_t nb, int64_t d,
                                                           int dtype, float neg_lambda, const float* __restrict__ P, int dp,
                                                           float neg_lambda_g, int offdiag, unsigned long long count, int s,
                                                           float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= s) return;
  const unsigned long long idx = mix64((unsigned long long)i) % count;
  const unsigned long long per = (unsigned long long)(offdiag ? nb - 1 : nb);
  const unsigned long long r = idx / per;
  unsigned long long c = idx - r * per;
  if (offdiag) c += (c >= r) ? 1ull : 0ull;
  float acc = 0.0f;
  for (int64_t k = lane; k < d; k += 64) {
    const float t = ld_elem(A, (int64_t)r * d + k, dtype) - ld_elem(B, (int64_t)c * d + k, dtype);
    acc = __builtin_fmaf(t, t, acc);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) {
    float v = expf(neg_lambda * acc);
    if (P) {
      float g = 0.0f;
      for (int e = 0; e < dp; ++e) { const float t = P[r * dp + e] - P[c * dp + e]; g = __builtin_fmaf(t, t, g); }
      v *= expf(neg_lambda_g * g);
    }
    out[i] = v;
  }
}

// one workgroup: the sample's order statistics r_lo and r_hi by radix select out of LDS -> the bracket; state reset.
// A thread owns every 1024th key and counts runs of equal bins in a register (similarities are concentrated: without
// that the first passes are 32768 atomics on one LDS word).
__global__ __launch_bounds__(1024) void bracket_bounds_kernel(const float* __restrict__ sample, int s, int r_lo, int r_hi,
                                                              unsigned long long rank, unsigned long long seg_cap, BracketState* st) {
  extern __shared__ uint32_t bkeys[];
  __shared__ unsigned int hist[256];
  __shared__ unsigned int cum[256];
  __shared__ uint32_t sh_prefix;
  __shared__ int sh_rank;
  const int tid = threadIdx.x;
  for (int i = tid; i < s; i += 1024) bkeys[i] = f2ord(sample[i]);
  for (int i = tid; i < kBracketSegs * kBracketPad; i += 1024) { st->cursor[i] = 0ull; st->below[i] = 0ull; }
  uint32_t found[2] = {0u, 0u};
  for (int which = 0; which < 2; ++which) {
    __syncthreads();
    if (tid == 0) { sh_prefix = 0u; sh_rank = which ? r_hi : r_lo; }
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (tid < 256) hist[tid] = 0u;
      __syncthreads();
      const uint32_t prefix = sh_prefix;
      const uint32_t himask = (shift == 24) ? 0u : (0xffffffffu << (shift + 8));
      uint32_t cur = 0xffffffffu, run = 0u;
      for (int i = tid; i < s; i += 1024) {        // interleaved ownership: conflict-free LDS reads
        const uint32_t k = bkeys[i];
        if ((k & himask) != (prefix & himask)) continue;
        const uint32_t bin = (k >> shift) & 255u;
        if (bin == cur) { ++run; continue; }
        if (run) atomicAdd(&hist[cur], run);
        cur = bin; run = 1u;
      }
      if (run) atomicAdd(&hist[cur], run);
      __syncthreads();
      if (tid < 256) cum[tid] = hist[tid];
      __syncthreads();
      for (int o = 1; o < 256; o <<= 1) {
        unsigned int v = 0u;
        if (tid < 256 && tid >= o) v = cum[tid - o];
        __syncthreads();
        if (tid < 256) cum[tid] += v;
        __syncthreads();
      }
      if (tid < 256) {
        const unsigned int r = (unsigned int)sh_rank, before = cum[tid] - hist[tid];
        if ((r >= before && r < cum[tid]) || (tid == 255 && r >= cum[255])) {
          sh_rank = (int)(r - before);
          sh_prefix = prefix | ((uint32_t)tid << shift);
        }
      }
      __syncthreads();
    }
    found[which] = sh_prefix;
  }
  if (tid == 0) {
    st->lo = ord2f(found[0]); st->hi = ord2f(found[1]);
    st->rank = rank; st->seg_cap = seg_cap; st->inside = 0ull; st->fail = 0u; st->pad = 0u;
  }
}

struct StatPartial { double s1, s2; float mn, mx; };

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}


// K: rows [row0, row0 + rows) of a matrix with n columns; the entry (row0 + li, j == row0 + li) is skipped (pass
// kNoDiagonal as row0 for data without a diagonal).  One wave per row, 16 bytes per lane when n % 4 == 0.
// part != NULL: the sweep also leaves the workgroup's statistic partial (sums around pivot[0], minimum, maximum) over the entries
// it visits — mmf_array_stats then reads its array once instead of twice.
__global__ __launch_bounds__(256) void bracket_sweep_kernel(const float* __restrict__ K, int64_t n, int64_t row0, int64_t rows,
                                                            BracketState* st, float* __restrict__ buf, StatPartial* __restrict__ part,
                                                            const float* __restrict__ pivot) {
  __shared__ float stage[4][1024];
  __shared__ unsigned long long wbelow[4];
  __shared__ StatPartial wstat[4];
  const double pv = part ? (double)pivot[0] : 0.0;
  double s1 = 0.0, s2 = 0.0;
  float smn = __builtin_huge_valf(), smx = -__builtin_huge_valf();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float lo = st->lo, hi = st->hi;
  const unsigned long long seg_cap = st->seg_cap;
  const int seg = (int)((blockIdx.x * 4u + (unsigned)w) % (unsigned)kBracketSegs);
  float* sbuf = buf + (size_t)seg * seg_cap;
  const unsigned long long lt = (1ull << lane) - 1ull;
  unsigned long long below = 0ull;
  int cnt = 0;                                   // wave-uniform: entries waiting in this wave's stage
  auto flush = [&]() {
    unsigned long long base = 0ull;
    if (lane == 0) base = atomicAdd(&st->cursor[seg * kBracketPad], (unsigned long long)cnt);
    base = ((unsigned long long)(uint32_t)__shfl((int)(base >> 32), 0) << 32) | (unsigned long long)(uint32_t)__shfl((int)(uint32_t)base, 0);
    for (int i = lane; i < cnt; i += 64)
      if (base + (unsigned long long)i < seg_cap) sbuf[base + (unsigned long long)i] = stage[w][i];
    cnt = 0;
  };
  auto offer = [&](float x, bool valid) {
    const bool in = valid && x >= lo && x <= hi;
    below += (valid && x < lo) ? 1ull : 0ull;
    if (part && valid) {
      const double dx = (double)x - pv;
      s1 += dx; s2 = __builtin_fma(dx, dx, s2);
      smn = fminf(smn, x); smx = fmaxf(smx, x);
    }
    const unsigned long long mask = __ballot(in);
    if (in) stage[w][cnt + __popcll(mask & lt)] = x;
    cnt += __popcll(mask);
  };
  const bool vec = ((n & 3) == 0) && ((reinterpret_cast<uintptr_t>(K) & 15) == 0);
  for (int64_t li = (int64_t)blockIdx.x * 4 + w; li < rows; li += (int64_t)gridDim.x * 4) {
    const float* row = K + li * n;
    const int64_t i = row0 + li;
    if (vec) {
      for (int64_t j0 = 0; j0 < n; j0 += 256) {          // wave-uniform trip count: the ballots see every lane
        const int64_t j = j0 + (int64_t)lane * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (j < n) v = *reinterpret_cast<const f32x4*>(row + j);
#pragma unroll
        for (int e = 0; e < 4; ++e) offer(v[e], j < n && j + e != i);
        if (cnt > 1024 - 256) flush();
      }
    } else {
      for (int64_t j0 = 0; j0 < n; j0 += 64) {
        const int64_t j = j0 + lane;
        offer(j < n ? row[j] : 0.0f, j < n && j != i);
        if (cnt > 1024 - 64) flush();
      }
    }
  }
  if (cnt > 0) flush();
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long hi32 = (unsigned long long)(uint32_t)__shfl_xor((int)(below >> 32), o);
    const unsigned long long lo32 = (unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)below, o);
    below += (hi32 << 32) | lo32;
  }
  if (part) {
    s1 = wave_sum(s1); s2 = wave_sum(s2);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { smn = fminf(smn, __shfl_xor(smn, o)); smx = fmaxf(smx, __shfl_xor(smx, o)); }
    if (lane == 0) wstat[w] = StatPartial{s1, s2, smn, smx};
  }
  if (lane == 0) wbelow[w] = below;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = wbelow[0] + wbelow[1] + wbelow[2] + wbelow[3];
    if (t) atomicAdd(&st->below[(blockIdx.x % (unsigned)kBracketSegs) * kBracketPad], t);
    if (part) {
      StatPartial r = wstat[0];
      for (int i = 1; i < 4; ++i) { r.s1 += wstat[i].s1; r.s2 += wstat[i].s2; r.mn = fminf(r.mn, wstat[i].mn); r.mx = fmaxf(r.mx, wstat[i].mx); }
      part[blockIdx.x] = r;
    }
  }
}

// does the bracket hold the wanted rank?  sets up the select inside the buffer
__global__ __launch_bounds__(64) void bracket_begin_kernel(BracketState* st) {
  const int t = threadIdx.x;
  unsigned long long c = st->cursor[t * kBracketPad], b = st->below[t * kBracketPad];
  const bool over = c > st->seg_cap;
  unsigned long long hi32, lo32;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    hi32 = (unsigned long long)(uint32_t)__shfl_xor((int)(c >> 32), o); lo32 = (unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)c, o);
    c += (hi32 << 32) | lo32;
    hi32 = (unsigned long long)(uint32_t)__shfl_xor((int)(b >> 32), o); lo32 = (unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)b, o);
    b += (hi32 << 32) | lo32;
  }
  const bool any_over = __any(over);
  for (int c = 0; c < kHistCopies; ++c)
    for (int i = t; i < 256; i += 64) st->sel.hist[c][i] = 0ull;
  if (t == 0) {
    const bool ok = !any_over && st->rank >= b && (st->rank - b) < c;
    st->fail = ok ? 0u : 1u;
    st->inside = c;
    st->sel.prefix = 0ull;
    st->sel.rank = ok ? (st->rank - b) : 0ull;
  }
}

// one radix pass over the buffered values: workgroup -> (segment, slice of it); runs of equal bins counted in a register
__global__ __launch_bounds__(256) void bracket_hist_kernel(const float* __restrict__ buf, BracketState* st, int shift) {
  __shared__ unsigned int lh[256];
  lh[threadIdx.x] = 0u;
  __syncthreads();
  if (st->fail == 0u) {
    const int seg = blockIdx.x % kBracketSegs, part = blockIdx.x / kBracketSegs, parts = gridDim.x / kBracketSegs;
    const unsigned long long count = st->cursor[seg * kBracketPad];
    const float* sbuf = buf + (size_t)seg * st->seg_cap;
    const uint32_t prefix = (uint32_t)st->sel.prefix;
    const uint32_t himask = (shift == 24) ? 0u : (0xffffffffu << (shift + 8));
    uint32_t cur = 0xffffffffu, run = 0u;
    for (unsigned long long i = (unsigned long long)part * 256 + threadIdx.x; i < count; i += (unsigned long long)parts * 256) {
      const uint32_t o = f2ord(sbuf[i]);
      if ((o & himask) != (prefix & himask)) continue;
      const uint32_t bin = (o >> shift) & 255u;
      if (bin == cur) { ++run; continue; }
      if (run) atomicAdd(&lh[cur], run);
      cur = bin; run = 1u;
    }
    if (run) atomicAdd(&lh[cur], run);
  }
  __syncthreads();
  if (lh[threadIdx.x]) atomicAdd(&st->sel.hist[blockIdx.x % kHistCopies][threadIdx.x], (unsigned long long)lh[threadIdx.x]);
}

__global__ void bracket_report_kernel(const BracketState* st, uint32_t* flag) { *flag = st->fail; }

size_t bracket_state_bytes() { return (sizeof(BracketState) + 255) & ~size_t(255); }
// buffer entries for a population of `count` (0: the one-sweep path is not used — small or enormous populations)
unsigned long long bracket_capacity(unsigned long long count) {
  if (getenv("MMF_MEDIAN_RADIX")) return 0ull;                   // A/B switch: always the four-pass select
  if (count < (1ull << 22)) return 0ull;
  unsigned long long cap = count / 20ull + 65536ull;             // 5 %: the bracket holds 3.3 % in expectation
  cap = (cap + kBracketSegs * 64 - 1) / (kBracketSegs * 64) * (kBracketSegs * 64);   // 64 segments of whole 256-byte lines
  if (cap * 4ull > (2ull << 30)) return 0ull;
  return cap;
}
size_t median_scratch_bytes(unsigned long long count) {
  return bracket_state_bytes() + kBracketSample * sizeof(float) + 256 + (size_t)bracket_capacity(count) * sizeof(float) + 256 +
         ((sizeof(MedianState) + 255) & ~size_t(255));
}

// out (device): the lower median of `count` values — torch.median's element (count - 1) / 2.
//   sampler(sample, s): fills s sample values (device);  sweep(consume): streams the whole population once through
//   consume(data, cols, row0, rows) (row0: global row of the first row for the diagonal skip, or kNoDiagonalRow).
// One host synchronisation (the bracket's verdict) when the one-sweep path is taken.
// stat_part / stat_pivot / stat_nparts (optional): the one-sweep path leaves statistic partials of the population there (one per
// workgroup of every sweep launch, *stat_nparts of them); *stat_nparts stays 0 when the sweep did not run.
int lower_median_of(unsigned long long count, const MedianSampler& sampler, const MedianSweep& sweep, float* out, void* scratch,
                    hipStream_t s, void* stat_part, const float* stat_pivot, int64_t* stat_nparts) {
  if (stat_nparts) *stat_nparts = 0;
  int64_t parts_used = 0;
  char* base = static_cast<char*>(scratch);
  BracketState* st = reinterpret_cast<BracketState*>(base);
  float* sample = reinterpret_cast<float*>(base + bracket_state_bytes());
  const unsigned long long cap = bracket_capacity(count);
  float* buf = sample + kBracketSample + 64;
  void* radix = reinterpret_cast<char*>(buf) + (((size_t)cap * sizeof(float) + 255) & ~size_t(255));
  const unsigned long long rank = (count - 1ull) / 2ull;
  if (cap != 0ull) {
    MMF_TRY(sampler(sample, kBracketSample));
    MMF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bracket_bounds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                kBracketSample * 4));
    hipLaunchKernelGGL(bracket_bounds_kernel, dim3(1), dim3(1024), kBracketSample * 4, s, sample, kBracketSample,
                       kBracketSample / 2 - kBracketHalfWidth, kBracketSample / 2 + kBracketHalfWidth, rank, cap / kBracketSegs, st);
    MMF_LAUNCH_CHECK();
    MMF_TRY(sweep([&](const float* data, int64_t cols, int64_t row0, int64_t rows) -> int {
      if (rows <= 0) return MMF_OK;
      int64_t grid = (rows + 3) / 4;
      if (grid > (stat_part ? 2040 : 2048)) grid = stat_part ? 2040 : 2048;     // a ragged last row adds one launch of one workgroup
      StatPartial* sp = stat_part ? reinterpret_cast<StatPartial*>(stat_part) + parts_used : nullptr;
      if (sp && parts_used + grid > 2048) { set_error("median sweep: more than 2048 statistic partials"); return MMF_E_INTERNAL; }
      hipLaunchKernelGGL(bracket_sweep_kernel, dim3((unsigned)grid), dim3(256), 0, s, data, cols, row0, rows, st, buf, sp, stat_pivot);
      MMF_LAUNCH_CHECK();
      if (sp) parts_used += grid;
      return MMF_OK;
    }));
    if (stat_nparts) *stat_nparts = parts_used;
    hipLaunchKernelGGL(bracket_begin_kernel, dim3(1), dim3(64), 0, s, st);
    MMF_LAUNCH_CHECK();
    MedianState* sel = &st->sel;
    for (int pass = 0; pass < 4; ++pass) {
      hipLaunchKernelGGL(bracket_hist_kernel, dim3(16 * kBracketSegs), dim3(256), 0, s, buf, st, 24 - 8 * pass);
      MMF_LAUNCH_CHECK();
      hipLaunchKernelGGL(median_pick_kernel, dim3(1), dim3(256), 0, s, sel, 24 - 8 * pass, out);
      MMF_LAUNCH_CHECK();
    }
    uint32_t h_fail = 1;
    MMF_HIP(hipMemcpyAsync(&h_fail, &st->fail, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    MMF_HIP(hipStreamSynchronize(s));
    if (h_fail == 0u) return MMF_OK;
  }
  // four-pass radix select over the whole population
  MMF_TRY(launch_median_begin_count(radix, count, s));
  for (int pass = 0; pass < 4; ++pass) {
    MMF_TRY(sweep([&](const float* data, int64_t cols, int64_t row0, int64_t rows) -> int {
      return launch_median_accumulate(data, cols, row0, rows, radix, pass, s);
    }));
    MMF_TRY(launch_median_next(radix, pass, out, s));
  }
  return MMF_OK;
}

int launch_sample_gather(const float* data, int64_t n_sq, unsigned long long count, float* sample, int s_count, hipStream_t s) {
  hipLaunchKernelGGL(sample_gather_kernel, dim3((unsigned)((s_count + 255) / 256)), dim3(256), 0, s, data, n_sq, count, s_count, sample);
  MMF_LAUNCH_CHECK();
  return MMF_OK;
}
int launch_sample_pairs(const void* A, const void* B, int64_t nb, int64_t d, int dtype, float lambda, const float* P, int dp,
                        float lambda_g, int offdiag, unsigned long long count, float* sample, int s_count, hipStream_t s) {
  hipLaunchKernelGGL(sample_pairs_kernel, dim3((unsigned)((s_count + 3) / 4)), dim3(256), 0, s, A, B, nb, d, dtype, -lambda, P, dp,
                     -lambda_g, offdiag, count, s_count, sample);
  MMF_LAUNCH_CHECK();
  return MMF_OK;
}

// flat array as rows of 4096 values plus one ragged row
static int sweep_flat(const float* v, int64_t count, const MedianConsume& consume) {
  const int64_t C = 4096, full = count / C, tail = count - full * C;
  if (full > 0) MMF_TRY(consume(v, C, kNoDiagonal, full));
  if (tail > 0) MMF_TRY(consume(v + full * C, tail, kNoDiagonal, 1));
  return MMF_OK;
}

// Lower median (torch.median semantics: element (count - 1) / 2 of the sorted values) of a flat array — the
// similarity statistics of preprocess_hypergraph.py:190-196, 259-265 and the edge-weight median of :885-897.
// The array is walked as rows of 4096 values (plus one ragged row) by the same histogram kernel; a row0 that no
// column index can reach switches the diagonal skip off.
int launch_lower_median(const float* v, int64_t count, float* out, void* scratch /* median_scratch_bytes(count) */, hipStream_t s) {
  return lower_median_of(
      (unsigned long long)count, [&](float* sample, int sc) { return launch_sample_gather(v, 0, (unsigned long long)count, sample, sc, s); },
      [&](const MedianConsume& consume) { return sweep_flat(v, count, consume); }, out, scratch, s);
}

// ------------------------------------------------------------------------------------------------
// mean / std / min / max of a flat f32 array in ONE pass (f64 accumulation around a pivot), deterministic:
// every workgroup writes its partial, one workgroup merges them in index order.
// Replaces: K.mean(), K.std(), K.min(), K.max() — four passes — at preprocess_hypergraph.py:190-195, 260-263.
// ------------------------------------------------------------------------------------------------
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

size_t array_stats_scratch_bytes(int64_t count) { return 2048 * sizeof(StatPartial) + 256 + median_scratch_bytes((unsigned long long)count); }

int launch_array_stats(const float* v, int64_t count, double* out, void* scratch, hipStream_t s) {
  StatPartial* part = reinterpret_cast<StatPartial*>(scratch);
  char* rest = reinterpret_cast<char*>(scratch) + 2048 * sizeof(StatPartial);
  float* med = reinterpret_cast<float*>(rest);
  void* mscratch = rest + 256;
  // large arrays: the median's one sweep also leaves the statistic partials (one read of the array instead of two)
  int64_t nparts = 0;
  MMF_TRY(lower_median_of(
      (unsigned long long)count, [&](float* sample, int sc) { return launch_sample_gather(v, 0, (unsigned long long)count, sample, sc, s); },
      [&](const MedianConsume& consume) { return sweep_flat(v, count, consume); }, med, mscratch, s, part, v, &nparts));
  if (nparts == 0) {
    int64_t grid = (count + 256 * 16 - 1) / (256 * 16);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(stats_partial_kernel, dim3((unsigned)grid), dim3(256), 0, s, v, count, part);
    MMF_LAUNCH_CHECK();
    nparts = grid;
  }
  MMF_TRY(launch_stats_finish(part, nparts, v, count, out, s));
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
  if (t == 1023) { *out_count = (int64_t)(base + part[1023]); row_off[n] = base + part[1023]; }
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

// The two halves of the above for a caller that sizes its outputs from the count: the counting pass leaves the rows'
// exclusive offsets (row_off[n + 1], row_off[n] = total), the fill pass takes them — K is read twice, not three times.
int launch_threshold_count(const float* K, int64_t n, float thr, unsigned long long* row_off, int64_t* out_count, uint32_t* row_cnt,
                           hipStream_t s) {
  const unsigned grid = (unsigned)((n + 3) / 4);
  hipLaunchKernelGGL(thr_count_kernel, dim3(grid), dim3(256), 0, s, K, n, n, thr, row_cnt);
  MMF_LAUNCH_CHECK();
  hipLaunchKernelGGL(thr_scan_kernel, dim3(1), dim3(1024), 0, s, row_cnt, n, row_off, out_count);
  MMF_LAUNCH_CHECK();
  return MMF_OK;
}
int launch_threshold_fill(const float* K, int64_t n, float thr, const unsigned long long* row_off, int64_t* ei, float* ew,
                          int64_t capacity, hipStream_t s) {
  const unsigned grid = (unsigned)((n + 3) / 4);
  hipLaunchKernelGGL(thr_fill_kernel, dim3(grid), dim3(256), 0, s, K, n, (int64_t)0, n, thr, row_off, ei, ei + capacity, ew, capacity);
  MMF_LAUNCH_CHECK();
  return MMF_OK;
}

}  // namespace mmf
