// mmf_host.h — host-side plumbing of libmmf_hg.so: error reporting, per-(device,stream)
// grow-only workspaces, launch checks, and the internal kernel-launcher prototypes.
#pragma once
#include <hip/hip_runtime.h>
#include <functional>
#include <stdint.h>
#include <stdio.h>

#include <string>
#include <vector>

#include "../../include/mmf_hg.h"

namespace mmf {

void set_error(const char* fmt, ...);

#define MMF_HIP(call)                                                                      \
  do {                                                                                     \
    hipError_t _e = (call);                                                                \
    if (_e != hipSuccess) {                                                                \
      mmf::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
      return MMF_E_HIP;                                                                    \
    }                                                                                      \
  } while (0)

#define MMF_LAUNCH_CHECK()                                                                 \
  do {                                                                                     \
    hipError_t _e = hipGetLastError();                                                     \
    if (_e != hipSuccess) {                                                                \
      mmf::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return MMF_E_HIP;                                                                    \
    }                                                                                      \
  } while (0)

#define MMF_TRY(expr)            \
  do {                           \
    int _rc = (expr);            \
    if (_rc != MMF_OK) return _rc; \
  } while (0)

// Bump allocator over one cached device buffer per (device, stream).
struct Workspace {
  char* base = nullptr;
  size_t cap = 0;
  size_t off = 0;
  void reset() { off = 0; }
  template <typename T>
  T* take(size_t count) {
    size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
    T* p = reinterpret_cast<T*>(base + off);
    off += bytes;
    return p;
  }
};
inline size_t ws_bytes(size_t count, size_t elem) { return (count * elem + 255) & ~size_t(255); }

// Returns a workspace of at least `bytes` for (device, stream); reallocates (after a stream sync)
// when it has to grow.
int get_workspace(int device, hipStream_t stream, size_t bytes, Workspace* out);

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

inline size_t dtype_size(int dt) { return dt == MMF_F32 ? 4 : 2; }

// ------------------------------------------------------------------------------------------------
// kernel launchers (one translation unit each)
// ------------------------------------------------------------------------------------------------

// mmf_prep.hip: canonical squared norms (or clamped norms for cosine) of every row.
// max_n (optional, device u32 holding float bits, zeroed by the caller) receives the largest n_i.
int launch_row_scalars(const void* X, int64_t n, int64_t d, int dtype, int metric, float* out,
                       uint32_t* max_n, hipStream_t s);

// candidate lists written by the scan kernels and read by the select kernel:
//   cnt[(row * lists + l)]            number of ids in list l of the row
//   ids[(row * lists + l) * cap + e]  LOCAL column index (0..m-1)
struct CandLists {
  uint32_t* cnt;
  uint32_t* ids;
  uint32_t* overflow;  // [n] nonzero when a list of the row overflowed
  int lists;           // lists per row = 2 * col_splits
  int cap;
  // 16-bit scan only (else nullptr): the entries' approximate keys and the row's error margin — with several
  // lists per row, select drops what the union of the lists proves irrelevant before the exact re-rank
  float* keys = nullptr;
  float* margin = nullptr;   // [n]
  int slot_ulp = 16;         // stored keys carry an id-slot number in their low mantissa bits: 16 (4 bits) or 32 (5 bits)
  // 16-bit scan only: per-row overflow lists (SpillSink, mmf_dev.h) — what the lane lists could not hold
  uint32_t* spill_cnt = nullptr;   // [n], zeroed before the first launch
  uint32_t* spill_ids = nullptr;   // [n][spill_cap], GLOBAL-mapped local column ids like `ids`
  int spill_cap = 0;
  int spill_stacks = 0;            // lists == 2: the row's two lanes fill its slots from both ends; spill_cnt[row] = front | back << 16
};

// Where a launch of the 16-bit scan sits inside a paneled scan (all zero: the whole problem in one launch).
struct ScanB16Panel {
  int list_base = 0;                                   // first of the row's list slots this launch writes
  uint32_t seg_len = 0, seg_stride = 0, id_off = 0;    // operand column i -> id_off + (i / seg_len) * seg_stride + i % seg_len
  int32_t* seed = nullptr;                             // [2][seed_stride]: best proven threshold / best dropped key per query
  int64_t seed_stride = 0;                             //   (ordered-int encoding), memset to 0x80 before the first launch
  int share = 0;                                       // more than one workgroup / launch scans each query
};

// mmf_prep.hip: the f32 operand image of mmf_scan_f32.hip — [prep_f32_rows(n)][prep_f32_dim(d)] floats, zero padded,
// de-interleaved inside every group of eight k (k0 k2 k4 k6 | k1 k3 k5 k7).  row_ids: optional gather of the rows.
int64_t prep_f32_rows(int64_t n);
int64_t prep_f32_dim(int64_t d);
size_t prep_f32_bytes(int64_t n, int64_t d);
int launch_prep_f32(const void* X, int64_t n, int64_t d, int dtype, const int32_t* row_ids, float* out, hipStream_t s);

struct ScanProblem {
  const void* X; int64_t n;       // query rows
  const void* Y; int64_t m;       // candidate rows (columns of the similarity matrix)
  const float* Xp = nullptr;      // f32 images (launch_prep_f32) of the rows to scan — gathered when row_ids is set —
  const float* Yp = nullptr;      //   and of the candidate rows: what the exact scan reads
  int64_t d;
  int dtype;
  int metric;
  float lambda;
  int kk;                         // entries to retain per query (k, +1 when self is excluded later)
  const float* rx;                // per-row scalar of X  (launch_row_scalars)
  const float* cy;                // per-row scalar of Y
  const int32_t* row_ids;         // optional: scan only these rows of X (fallback), else nullptr
  int64_t n_rows;                 // number of rows to scan (== n when row_ids == nullptr)
  int col_splits;
  const float* floor_key = nullptr;     // exact scan only: per query, offer only what ranks strictly after (floor_key, floor_id)
  const uint32_t* floor_id = nullptr;   //   in the total order — the later passes of a call with k + self > 44
};

// mmf_scan_f32.hip: exact scan on v_mfma_f32_32x32x2_f32 (any d, f32/bf16/f16 inputs).
int scan_f32_cap(int kk);  // list capacity the kernel will use for kk (0 = unsupported)
int launch_scan_f32(const ScanProblem& p, const CandLists& L, hipStream_t s, int* grid_out);

// mmf_select.hip: canonical keys of the candidates, self dropped, top-k by (key desc, id asc).
struct SelectProblem {
  const void* X; int64_t n; const void* Y; int64_t m; int64_t d; int dtype; int metric; float lambda;
  int k; int exclude_self; int64_t row_offset; int64_t col_offset;
  const float* rx; const float* cy;
  const int32_t* row_ids; int64_t n_rows;
  int64_t* out_idx; float* out_val;
  int32_t* fail_rows; uint32_t* fail_count;   // rows whose lists overflowed (or came up short)
  uint32_t* cand_total;                        // optional accumulated candidate count
  bool two_pass = false;   // rows with overflow-list entries are handled by a second launch that has LDS room for them
  int out_stride = 0, out_off = 0;       // out_idx / out_val rows are out_stride wide (0: k) and this call fills columns out_off .. out_off + k
  float* floor_key_out = nullptr;        // optional: key and LOCAL column id of the last entry emitted per row (the next pass's floor)
  uint32_t* floor_id_out = nullptr;
  void* order_scratch = nullptr;   // select_order_bytes(n_rows): that second launch walks its rows ordered by their smallest candidate id
  const int32_t* perm = nullptr;   // optional (row_ids must be null): list position -> row of X, a permutation of 0 .. n_rows - 1 — the scan
                                   // took its queries in that order (mmf_order.hip); X, rx, the outputs and fail_rows stay in row order
};
size_t select_order_bytes(int64_t n);
// mmf_order.hip: the order in which the 16-bit scan takes its query rows (near-duplicate rows next to each other)
size_t query_order_bytes(int64_t n);
int query_order_pivots();
int query_order_last(int32_t* out_host, int64_t n);
void query_order_forget();   // the recorded permutation is gone (a new fast-path call has begun, or the workspaces were released)
int launch_query_order_probe(const uint16_t* ZQ, const float* q_zn, int64_t n, int dp, bool f16, void* scratch, int64_t* near, hipStream_t s);
int launch_query_order_apply(const uint16_t* ZQ, const float* q_zn, const float* q_rn, const float* q_un, int64_t n, int64_t n_pad,
                             int dp, bool f16, void* scratch, uint16_t* Zo, float* zno, float* rno, float* uno, const int32_t** perm,
                             hipStream_t s);
int launch_select(const SelectProblem& p, const CandLists& L, hipStream_t s);
// exact top-k of a few rows (p.row_ids) against every column, no candidate lists; keys: p.n_rows * p.m floats
int launch_rows_exact(const SelectProblem& p, float* keys, hipStream_t s);
int launch_topk_merge(const int64_t* ia, const float* va, const int64_t* ib, const float* vb,
                      int64_t n, int k, int64_t* io, float* vo, hipStream_t s);
int launch_edge_cosine(const void* X, int64_t n, int64_t d, int dtype, const int64_t* ei, int64_t E,
                       float* out, hipStream_t s);

// mmf_dense.hip
// Xp / Yp: f32 images of X and Y (launch_prep_f32; unused — may be null — for d <= 8 and MMF_RBF_DIRECT)
bool sim_dense_needs_images(int64_t d, int metric);
int launch_sim_dense(const void* X, int64_t n, const void* Y, int64_t m, int64_t d, int dtype,
                     int metric, float lambda, const float* rx, const float* cy, const float* Xp, const float* Yp,
                     float* out, hipStream_t s);
// rows [row0, row0 + rows) of the combined n x n similarity; out [rows, n].  Fp: f32 image of all n rows of F.
int launch_sim_dense_combined(const float* Fp, const float* P, int64_t n, int64_t d, int64_t dp,
                              float lambda_h, float lambda_g, const float* nf, int64_t row0, int64_t rows, float* out,
                              hipStream_t s);

// mmf_edges.hip
// Lower medians (torch.median: element (count - 1) / 2).  A population of 4 M values or more is first tried in ONE sweep
// (sampled bracket, exact counts, select among the ~3 % inside the bracket; one host sync for the verdict); the four-pass
// radix select is the fallback and the small-size path.  MMF_MEDIAN_RADIX=1 forces the latter.
using MedianConsume = std::function<int(const float* data, int64_t cols, int64_t row0, int64_t rows)>;
using MedianSweep = std::function<int(const MedianConsume&)>;      // streams the whole population once through `consume`
using MedianSampler = std::function<int(float* sample, int s)>;   // s values at hashed positions (device)
size_t median_scratch_bytes(unsigned long long count);
int64_t median_no_diagonal_row();                                  // row0 for data without a diagonal to skip
int lower_median_of(unsigned long long count, const MedianSampler& sampler, const MedianSweep& sweep, float* out, void* scratch,
                    hipStream_t s, void* stat_part = nullptr, const float* stat_pivot = nullptr, int64_t* stat_nparts = nullptr);
int launch_sample_gather(const float* data, int64_t n_sq, unsigned long long count, float* sample, int s_count, hipStream_t s);
int launch_sample_pairs(const void* A, const void* B, int64_t nb, int64_t d, int dtype, float lambda, const float* P, int dp,
                        float lambda_g, int offdiag, unsigned long long count, float* sample, int s_count, hipStream_t s);
int launch_offdiag_lower_median(const float* K, int64_t n, float* out, void* scratch /* median_scratch_bytes(n (n - 1)) */,
                                hipStream_t s);
// the radix select in pieces (matrices recomputed panel by panel): begin; for pass 0..3 { accumulate panels; next }
int launch_median_accumulate(const float* K, int64_t n, int64_t row0, int64_t rows, void* state, int pass, hipStream_t s);
int launch_median_next(void* state, int pass, float* out, hipStream_t s);
int launch_median_begin_count(void* state, unsigned long long count, hipStream_t s);
// lower median of a flat array (scratch: median_scratch_bytes(count)); mean / std / min / max / median of a flat array
int launch_lower_median(const float* v, int64_t count, float* out, void* scratch, hipStream_t s);
int launch_stats_finish(const void* part, int64_t nparts, const float* pivot, int64_t count, double* out, hipStream_t s);
int launch_stats_set_median(const float* med, double* out, hipStream_t s);
size_t stat_partial_bytes();
// mmf_direct.hip: register-tiled direct-difference RBF with optional per-workgroup statistic partials
int64_t rbf_direct_blocks(int64_t n, int64_t m);
int launch_rbf_direct_pivot(const void* X, const void* Y, int64_t d, int dtype, float lambda, float* pivot, hipStream_t s);
int launch_rbf_direct(const void* X, int64_t n, const void* Y, int64_t m, int64_t d, int dtype, float lambda, float* out,
                      void* part, const float* pivot, hipStream_t s);
size_t array_stats_scratch_bytes(int64_t count);
int launch_array_stats(const float* v, int64_t count, double* out /*[5] device*/, void* scratch, hipStream_t s);
int launch_threshold_edges_panel(const float* K, int64_t n, int64_t row0, int64_t rows, float thr, int64_t* ei_row,
                                 int64_t* ei_col, float* ew, int64_t capacity, int64_t* out_count, uint32_t* scratch,
                                 size_t scratch_u32, hipStream_t s);
int launch_threshold_edges(const float* K, int64_t n, float thr, int64_t* ei, float* ew,
                           int64_t capacity, int64_t* out_count, uint32_t* scratch, size_t scratch_u32,
                           hipStream_t s);

int launch_threshold_count(const float* K, int64_t n, float thr, unsigned long long* row_off, int64_t* out_count, uint32_t* row_cnt,
                           hipStream_t s);
int launch_threshold_fill(const float* K, int64_t n, float thr, const unsigned long long* row_off, int64_t* ei, float* ew,
                          int64_t capacity, hipStream_t s);

// mmf_segments.hip: cluster-shaped steps (labels -> members, per-cluster means, cliques, k-NN pair dedup)
int segment_max_segments();
size_t segment_sort_scratch_bytes(int64_t n, int64_t S);
int launch_segment_sort(const int64_t* labels, int64_t n, int64_t S, int64_t* counts, int64_t* offsets, int64_t* order,
                        void* scratch, uint32_t* bad, hipStream_t s);
int launch_segment_mean(const float* X, int64_t d, const int64_t* order, const int64_t* offsets, int64_t S, float* out, hipStream_t s);
size_t segment_offdiag_scratch_bytes(int64_t n);
int launch_segment_offdiag_mean(const float* K, int64_t n, const int64_t* order, const int64_t* offsets, int64_t S,
                                double* out_mean, void* scratch, hipStream_t s);
size_t clique_scratch_bytes(int64_t n, int64_t S);
int launch_clique_pairs(const int64_t* order, const int64_t* offsets, int64_t n, int64_t S, int64_t* lo, int64_t* hi,
                        int64_t capacity, int64_t* out_count, void* scratch, hipStream_t s);
// mmf_kmeans.hip: scikit-learn's KMeans fit, decision for decision, all restarts in lockstep (host-synchronous)
size_t kmeans_scratch_bytes(int64_t n, int64_t d, int64_t k, int64_t n_init, int trials);
int launch_kmeans_fit(const float* X, int64_t n, int64_t d, int64_t k, int64_t n_init, int trials, const int64_t* first_h,
                      const double* u_h, int max_iter, double tol, int64_t* out_labels, float* out_centres, int64_t* out_seeds,
                      double* info_h, void* scratch, hipStream_t s);
int launch_knn_pairs(const int64_t* nbr, int64_t n, int k, const int64_t* labels, int64_t* lo, int64_t* hi, int64_t* out_count,
                     hipStream_t s);

}  // namespace mmf
