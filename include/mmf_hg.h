/*
 * mmf_hg.h — C ABI of libmmf_hg.so: MI355X (gfx950) hypergraph-construction hot path.
 *
 * The reference (zz9tf/multimodal-fusion) is pure Python and has no FFI of its own; its
 * boundary for this path is the set of Python functions re-exported by
 *   build_hypergraph/__init__.py:5-46  and  hypergraph/build_hypergraph/__init__.py:5-19.
 * Each entry point below names the reference code it replaces (paths relative to the
 * reference root).  The Python mirror of those functions lives in
 * multimodal-fusion_amd/build_hypergraph/ and binds this ABI with ctypes
 * (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - All pointers are raw DEVICE addresses (tensor.data_ptr()) unless a parameter says host.
 *   - Matrices are row-major and contiguous.  Indices are int64, scores are f32.
 *   - Every call enqueues on `hip_stream` (a hipStream_t, NULL = default stream) of device
 *     `device_id`.  Calls are asynchronous w.r.t. the host except where noted.
 *   - Return value: MMF_OK (0) or a negative MMF_E_* code.  Never throws, never aborts;
 *     mmf_last_error() gives the message of the calling thread's last failure.
 *   - There is NO CPU implementation behind this ABI.  device_id < 0 is rejected with
 *     MMF_E_UNSUPPORTED: the CPU restatement lives in oracle/ and is test infrastructure only.
 *
 * Canonical arithmetic (what "bit-exact" means; DESIGN.md §3 has the full statement)
 *   chain(a,b)   = acc = +0.0f; for k = 0..d-1: acc = fmaf(a[k], b[k], acc)      (k ascending)
 *   n_i          = chain(x_i, x_i);   dot_ij = chain(x_i, y_j)
 *   sq_ij        = (n_i + n_j) - 2*dot_ij           (similarity_kernel.py:49, same op order)
 *   MMF_DOT      key = val = dot_ij
 *   MMF_COSINE   key = val = dot_ij / (max(sqrtf(n_i),1e-8f) * max(sqrtf(n_j),1e-8f))
 *   MMF_NEG_SQ_L2 key = val = -sq_ij
 *   MMF_RBF      key = (-lambda)*sq_ij, val = expf(key)  (similarity_kernel.py:52)
 *   Ranking: key descending, then GLOBAL column id ascending; self (global col id == global
 *   row id) dropped by identity when exclude_self != 0.
 *   bf16 / f16 inputs are upcast to f32 exactly and then follow the same definition.
 */
#ifndef MMF_HG_H
#define MMF_HG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMF_ABI_VERSION 3

/* return codes */
#define MMF_OK             0
#define MMF_E_INVALID     (-1)  /* bad shape / argument          -> Python ValueError   */
#define MMF_E_UNSUPPORTED (-2)  /* valid but not implemented     -> Python RuntimeError */
#define MMF_E_HIP         (-3)  /* HIP runtime failure           -> Python RuntimeError */
#define MMF_E_NOMEM       (-4)  /* workspace allocation failed   -> Python RuntimeError */
#define MMF_E_INTERNAL    (-5)  /* invariant broken (a bug)      -> Python RuntimeError */

/* metric */
#define MMF_DOT        0
#define MMF_COSINE     1
#define MMF_NEG_SQ_L2  2
#define MMF_RBF        3
#define MMF_RBF_DIRECT 4  /* dense only: exp(-lambda * sum_k (a_k-b_k)^2), preprocess_hypergraph.py:254-256 */

/* element type of X / Y */
#define MMF_F32  0
#define MMF_BF16 1
#define MMF_F16  2

/* candidate-generation precision for mmf_simtopk_ex.  The RESULT is identical in every mode
 * (final keys are always the canonical f32 chain); the mode only picks which MFMA pipe scans
 * the N x M pairs. */
#define MMF_PREC_AUTO  0  /* FAST when the 16-bit scan supports the shape (d <= 1024 and k + self <= 20, or
                             d <= 512 and k + self <= 44), else EXACT (any d, any k: k + self > 44
                             takes ceil(k / (44 - self)) passes, each ranked after the one before)  */
#define MMF_PREC_EXACT 1  /* v_mfma_f32_32x32x2_f32 scan, canonical keys in-kernel           */
#define MMF_PREC_FAST  2  /* f16 MFMA scan (rows scaled by an exact power of two) with a proven error
                             margin + exact f32 re-rank; columns inside a row's margin that do not fit its
                             lists go to a per-row overflow list, rows that exhaust it are rescanned exactly */
#define MMF_PREC_FAST_BF16 3 /* same with bf16 operands: 8x larger rounding residual, wider margin    */

/* order in which the 16-bit scan takes its query rows (mmf_simtopk_opts.query_order).  A scan wave serves 32 consecutive
 * queries and leaves its matrix-core loop whenever any of them has a column inside its margin, so on near-duplicate data
 * (patch embeddings) it pays to put near-duplicate rows next to each other.  The order is an internal matter of the scan:
 * outputs stay in the caller's row order and their bits do not depend on it. */
#define MMF_QUERY_ORDER_AUTO 0 /* n, m >= 32768: probe (a sample of the rows against 128 pivot rows, one small matrix-core launch and one
                                  host synchronisation) and reorder when the rows have near-duplicates; smaller: row order   */
#define MMF_QUERY_ORDER_OFF  1 /* the caller's row order                                                                     */
#define MMF_QUERY_ORDER_ON   2 /* always reorder (tests)                                                                      */

int         mmf_version(void);
const char* mmf_last_error(void);
/* diagnostics: the scan position -> row permutation of the most recent call that reordered its n queries (host buffer, n entries);
 * valid until the next fast-path call or mmf_release_workspaces.  MMF_E_INVALID when there is none. */
int         mmf_debug_query_order(int32_t* perm_host, int64_t n);

/*
 * Fused similarity + per-row top-k; the N x M matrix never reaches HBM.
 * Replaces: torch.mm + elementwise passes (build_hypergraph/similarity_kernel.py:43-52),
 *           sklearn NearestNeighbors(k+1).kneighbors + "drop column 0"
 *           (build_hypergraph/preprocess_hypergraph.py:379-388) when metric = MMF_NEG_SQ_L2,
 *           and the per-row Python loop of compute_wsi_tma_similarity (:250-257) when only the
 *           best matches per row are wanted.
 * X:[n,d], Y:[m,d] (Y == NULL -> Y = X, m = n).  in_dtype: MMF_F32 / MMF_BF16 / MMF_F16.
 * exclude_self: drop the column whose global id (col_offset + j) equals row_offset + i.
 * out_idx:[n,k] int64 GLOBAL column ids, out_val:[n,k] f32; both sorted (key desc, id asc).
 * Errors: k < 1, k > (number of admissible columns), d < 1, n < 0, lambda <= 0 for MMF_RBF
 *         -> MMF_E_INVALID.   n == 0 is a no-op.
 * Host-synchronous once per call (reads back one fallback counter).
 */
int mmf_simtopk(const void* X, int64_t n, const void* Y, int64_t m, int64_t d,
                int in_dtype, int metric, float lambda, int k, int exclude_self,
                int64_t row_offset, int64_t col_offset,
                int64_t* out_idx, float* out_val,
                int device_id, void* hip_stream);

/* Same, with the knobs the bench and the tests need. */
typedef struct mmf_simtopk_opts {
  int      precision;      /* MMF_PREC_*                                                        */
  int      profile;        /* 1: bracket the scan kernel with HIP events on hip_stream           */
  int      col_splits;     /* 0 = auto; >0 forces the number of column ranges per row block      */
  int      query_order;    /* MMF_QUERY_ORDER_*: order in which the 16-bit scan takes the query rows (results do not depend on it) */
  void*    select_wait_event; /* optional hipEvent_t: the stream waits for it after the 16-bit scan and
                                 before anything reads the f32 rows of X / Y (overlapped all-gather)    */
} mmf_simtopk_opts;

typedef struct mmf_simtopk_stats {
  float    scan_ms;        /* duration of the scan (candidate) kernel, valid when profile = 1    */
  float    prep_ms;        /* norms / bf16 conversion kernels                                    */
  float    rerank_ms;      /* exact re-rank + select kernel                                      */
  float    fallback_ms;    /* exact rescans of overflowed rows (0 when none)                     */
  int64_t  candidates;     /* total candidates handed to the re-rank                            */
  int64_t  fallback_rows;  /* rows whose candidate list overflowed and were rescanned exactly    */
  int      precision_used; /* MMF_PREC_EXACT or MMF_PREC_FAST                                    */
  int      col_splits;     /* column ranges per row block actually used                         */
  int      scan_grid;      /* workgroups launched by the scan kernel                             */
  float    scan_wait_ms;   /* paneled scan, profile = 1: part of scan_ms the stream spent waiting for panels'
                              ready_events (exposed exchange time), 0 otherwise                   */
  int64_t  overflow_rows;  /* fallback rows whose candidate list overflowed (near-ties beyond capacity) */
  int64_t  short_rows;     /* fallback rows whose lists held fewer than k admissible candidates   */
  int64_t  near_rows;      /* estimated query rows within cosine 0.965 of one of 128 pivot rows other than themselves (-1: not probed) */
  float    order_ms;       /* profile = 1: pivot keys + sort + gather of the query order (0 when not tried)      */
  int      query_order;    /* 1: the scan took the queries with near-duplicate rows next to each other           */
} mmf_simtopk_stats;

int mmf_simtopk_ex(const void* X, int64_t n, const void* Y, int64_t m, int64_t d,
                   int in_dtype, int metric, float lambda, int k, int exclude_self,
                   int64_t row_offset, int64_t col_offset,
                   int64_t* out_idx, float* out_val,
                   const mmf_simtopk_opts* opts /* NULL = defaults */,
                   mmf_simtopk_stats* stats /* host, may be NULL */,
                   int device_id, void* hip_stream);

/*
 * Phase API of the fast path, for the row-sharded multi-GPU driver: every rank prepares
 * the 16-bit operands of ITS rows once, ranks exchange them (half the bytes of the f32 rows), and the
 * scan runs on prepared operands while the f32 rows — needed only by the exact re-rank — are still
 * arriving.  Results are bit-identical to mmf_simtopk.  No reference counterpart.
 *
 *   mmf_padded_dim(d)      padded feature dim DP of the 16-bit operands (0: d unsupported, use mmf_simtopk)
 *   mmf_row_scalars        scal[i] = canonical n_i (clamped norm for MMF_COSINE); *max_sq_norm is raised
 *                          atomically to max n_i (caller zeroes it; may be NULL)
 *   mmf_prep_rows          Z[n_pad, DP] = round_16(u * 2^e) with e from *max_sq_norm (the GLOBAL maximum:
 *                          all-reduce it first; ignored for MMF_COSINE), rows n..n_pad zero with cb = -inf;
 *                          zn / rn / un / cb [n_pad]; maxima[4] raised atomically (caller zeroes; all-reduce MAX)
 *   mmf_simtopk_prepared   X:[n,d], Y:[m,d] are the ORIGINAL rows (read by the re-rank only);
 *                          q / c are the prepared query / candidate sides.  c->Z and c->cb hold m_pad rows
 *                          (multiple of 256, >= m; padding rows zero / -inf); q->Z, q->zn, q->rn, q->un must be
 *                          readable for n rounded up to 256 rows (e.g. views into the candidate side
 *                          allocated with 256 rows of slack).  maxima[4] = global maxima of the candidate side.
 */
typedef struct mmf_prepared_side {
  const void*  Z;
  const float* scal;
  const float* zn;
  const float* rn;
  const float* un;
  const float* cb;
} mmf_prepared_side;

int64_t mmf_padded_dim(int64_t d);
/* 1 when the 16-bit scan (MMF_PREC_FAST / the phase API) handles feature dim d with k neighbours
 * (+ self when exclude_self), else 0 (such shapes run on the exact scan under MMF_PREC_AUTO). */
int mmf_fast_scan_supported(int64_t d, int k, int exclude_self);
int mmf_row_scalars(const void* X, int64_t n, int64_t d, int in_dtype, int metric, float* scal,
                    float* max_sq_norm, int device_id, void* hip_stream);
int mmf_prep_rows(const void* X, int64_t n, int64_t d, int in_dtype, int metric, int operand,
                  const float* scal, const float* max_sq_norm, void* Z, int64_t n_pad,
                  float* zn, float* rn, float* un, float* cb, float* maxima,
                  int device_id, void* hip_stream);
int mmf_simtopk_prepared(const void* X, int64_t n, const void* Y, int64_t m, int64_t d,
                         int in_dtype, int metric, float lambda, int k, int exclude_self,
                         int64_t row_offset, int64_t col_offset,
                         const mmf_prepared_side* q, const mmf_prepared_side* c, int64_t m_pad,
                         const float* maxima, int operand,
                         int64_t* out_idx, float* out_val,
                         const mmf_simtopk_opts* opts, mmf_simtopk_stats* stats,
                         int device_id, void* hip_stream);

/*
 * Paneled variant of mmf_simtopk_prepared: the candidate-side 16-bit operands arrive as n_panels separate
 * blocks (e.g. the chunks of a pipelined all-gather) and each block is scanned by its own launch as soon as
 * its `ready_event` (a hipEvent_t, or NULL) has fired, so the exchange of block p+1 overlaps the scan of
 * block p.  Launches hand their per-row thresholds on to the next, so later panels start warm.
 *   panel column i  <->  column  id_base + (i / seg_len) * seg_stride + i % seg_len  of Y
 *   (seg_len == 0: id_base + i).  With P ranks owning `rows` rows each and chunk c of S holding rows
 *   [c*rows/S, (c+1)*rows/S) of every rank in rank order: seg_len = rows/S, seg_stride = rows, id_base = c*rows/S.
 *   Z  [m_pad, padded_dim] rows past m zero;  cb [m_pad] entries past m -inf;  m_pad % 256 == 0, plus 256 rows /
 *   entries of readable slack behind each.
 * c_scal [m]: candidate-side row scalars in Y's (global) row order.  Everything else as mmf_simtopk_prepared.
 * Result bits are those of mmf_simtopk on the same X, Y.  No reference counterpart.
 */
typedef struct mmf_panel {
  const void*  Z;
  const float* cb;
  int64_t m;
  int64_t m_pad;
  int64_t seg_len;
  int64_t seg_stride;
  int64_t id_base;
  void*   ready_event;
} mmf_panel;

int mmf_simtopk_panels(const void* X, int64_t n, const void* Y, int64_t m, int64_t d,
                       int in_dtype, int metric, float lambda, int k, int exclude_self,
                       int64_t row_offset, int64_t col_offset,
                       const mmf_prepared_side* q, const float* c_scal,
                       const mmf_panel* panels, int n_panels,
                       const float* maxima, int operand,
                       int64_t* out_idx, float* out_val,
                       const mmf_simtopk_opts* opts, mmf_simtopk_stats* stats,
                       int device_id, void* hip_stream);

/*
 * Merge two sorted [n,k] partial results into one (column-panel streaming, cross-shard merges).
 * Order and tie-break as above; an id present in both inputs is kept once.  ids < 0 are padding.
 * No reference counterpart (the reference is single-device).
 */
int mmf_topk_merge(const int64_t* ia, const float* va, const int64_t* ib, const float* vb,
                   int64_t n, int k, int64_t* io, float* vo, int device_id, void* hip_stream);

/*
 * Edge weights for an explicit edge list: w_e = max(0, cos(x_i, x_j)) with the canonical cosine.
 * Replaces: the per-edge F.cosine_similarity(...).item() loop,
 *           build_hypergraph/preprocess_hypergraph.py:414-420.
 * edge_index:[2,E] int64 (row 0 = i, row 1 = j), out_w:[E] f32.  ids outside [0,n) -> MMF_E_INVALID
 * is NOT checked on device; the Python mirror validates.
 */
int mmf_edge_cosine(const void* X, int64_t n, int64_t d, int in_dtype,
                    const int64_t* edge_index, int64_t E, float* out_w,
                    int device_id, void* hip_stream);

/*
 * Dense [n,m] f32 similarity for the small-N reference signatures.
 * Replaces: compute_morphological_similarity / compute_spatial_similarity
 *           (build_hypergraph/similarity_kernel.py:17-54, 57-86; same code in
 *           hypergraph/build_hypergraph/similarity_kernel.py) with metric = MMF_RBF, and
 *           compute_wsi_tma_similarity (build_hypergraph/preprocess_hypergraph.py:248-257) with
 *           metric = MMF_RBF_DIRECT.
 * Y == NULL -> Y = X.  out:[n,m] f32.  For MMF_RBF any lambda is accepted (as the reference).
 */
int mmf_sim_dense(const void* X, int64_t n, const void* Y, int64_t m, int64_t d,
                  int in_dtype, int metric, float lambda, float* out,
                  int device_id, void* hip_stream);

/*
 * mmf_sim_dense plus the five statistics the reference takes of such a matrix — mean, std (unbiased), min, max,
 * lower median, as device double[5] — replacing the `.mean()/.std()/.min()/.max()/.median()` passes of
 * compute_wsi_tma_similarity (build_hypergraph/preprocess_hypergraph.py:259-265).
 * MMF_RBF_DIRECT: the sums, minimum and maximum come out of the epilogue of the kernel that forms S (no extra pass);
 * the median is one more sweep over S (see "Lower medians" below).  out == NULL (MMF_RBF_DIRECT only): S is never
 * stored — its rows are recomputed in panels of `panel_rows` rows (0 = about 1 GiB), once for the statistics and the
 * median together (four more times if the median's one-sweep path has to fall back).
 * Other metrics: mmf_sim_dense followed by mmf_array_stats.   n, m >= 1.
 */
int mmf_sim_dense_stats(const void* X, int64_t n, const void* Y, int64_t m, int64_t d,
                        int in_dtype, int metric, float lambda, float* out, double* out_stats,
                        int64_t panel_rows, int device_id, void* hip_stream);

/*
 * Dense combined similarity K = K_h * K_g in one pass (no K_h / K_g temporaries).
 * Replaces: compute_combined_similarity, build_hypergraph/similarity_kernel.py:88-124.
 * F:[n,d] features, P:[n,dp] positions (dp = 2 or 3, any dp >= 1 accepted), both f32.
 * out:[n,n] f32 = exp(-lambda_h*sq_h) * exp(-lambda_g*sq_g), each factor rounded to f32 first
 * (similarity_kernel.py:122).
 */
int mmf_sim_dense_combined(const float* F, const float* P, int64_t n, int64_t d, int64_t dp,
                           float lambda_h, float lambda_g, float* out,
                           int device_id, void* hip_stream);

/*
 * Threshold edge builder over a dense [n,n] similarity already in HBM.
 * Replaces: build_weighted_hypergraph's median + double loop,
 *           build_hypergraph/similarity_kernel.py:183-202 (row-major order, self-loops kept).
 *   mmf_offdiag_lower_median: lower median (torch.median semantics) of the n(n-1) off-diagonal
 *                             entries; result written to *out_median (device f32).
 * Lower medians (here, mmf_lower_median, mmf_array_stats, mmf_sim_dense_stats, mmf_combined_offdiag_median): a
 * population of 2^22 values or more is first tried in ONE sweep — 32768 entries at hashed positions give a bracket
 * [lo, hi] around the median, the sweep counts the entries below lo exactly and buffers the ~3 % inside the bracket,
 * and the median is selected among those; the answer is exact (the sample only chooses where to look).  If the
 * bracket misses the median's rank or the buffer overflows (many equal values), the four-pass radix select over the
 * whole population runs instead; smaller populations always take it.  The one-sweep path synchronises the stream
 * once (the bracket's verdict comes back to the host).  MMF_MEDIAN_RADIX=1 in the environment forces the radix path.
 *   mmf_threshold_edges:      keep (i,j) with K[i,j] >= threshold, row-major; writes at most
 *                             `capacity` edges, always writes the true count to *out_count
 *                             (device int64).  edge_index:[2,capacity] int64, edge_w:[capacity].
 */
int mmf_offdiag_lower_median(const float* K, int64_t n, float* out_median,
                             int device_id, void* hip_stream);
int mmf_threshold_edges(const float* K, int64_t n, float threshold,
                        int64_t* edge_index, float* edge_w, int64_t capacity,
                        int64_t* out_count, int device_id, void* hip_stream);
/*
 * The same builder in two calls for a caller that sizes edge_index from the count (a torch front end): the counting pass
 * leaves the rows' exclusive offsets (row_offsets[n + 1] device uint64, row_offsets[n] = *out_count), the fill pass takes
 * them and writes edge_index [2, capacity] / edge_w [capacity] — K is read twice in all (mmf_threshold_edges called with
 * capacity 0 and then again reads it three times).  K and threshold must be those of the counting call.
 */
int mmf_threshold_edges_count(const float* K, int64_t n, float threshold, uint64_t* row_offsets, int64_t* out_count,
                              int device_id, void* hip_stream);
int mmf_threshold_edges_fill(const float* K, int64_t n, float threshold, const uint64_t* row_offsets, int64_t* edge_index,
                             float* edge_w, int64_t capacity, int device_id, void* hip_stream);

/*
 * Order statistics of a flat f32 array already in HBM (any dense similarity matrix, or a vector of edge weights).
 *   mmf_lower_median: element (count-1)/2 of the sorted values = torch.median — the edge-weight median filter of
 *                     rebuild_hypergraph_from_similarity, build_hypergraph/preprocess_hypergraph.py:885-897.
 *   mmf_array_stats:  out_stats (device double[5]) = mean, std (unbiased, torch.std), min, max, lower median —
 *                     the similarity statistics of preprocess_hypergraph.py:186-197, 259-265, 849-855 — in one
 *                     reduction pass (f64 accumulation around a pivot, fixed merge order: bit-reproducible) plus
 *                     the median's sweep, instead of five torch reductions and a full sort.
 * count < 1 -> MMF_E_INVALID.
 */
int mmf_lower_median(const float* v, int64_t count, float* out_median, int device_id, void* hip_stream);
int mmf_array_stats(const float* v, int64_t count, double* out_stats, int device_id, void* hip_stream);

/*
 * The same two steps for an N whose combined similarity K = K_h * K_g ([n,n] f32) does not fit in
 * memory: K is recomputed from (F, P) in row panels of `panel_rows` rows (0 = about 1 GiB per panel) — one sweep
 * for the median (four when its one-sweep path falls back), one sweep per call of the edge builder (capacity 0
 * counts, then fill).
 * Results are those of mmf_sim_dense_combined + mmf_offdiag_lower_median / mmf_threshold_edges, bit for bit.
 */
int mmf_combined_offdiag_median(const float* F, const float* P, int64_t n, int64_t d, int64_t dp,
                                float lambda_h, float lambda_g, int64_t panel_rows, float* out_median,
                                int device_id, void* hip_stream);
int mmf_combined_threshold_edges(const float* F, const float* P, int64_t n, int64_t d, int64_t dp,
                                 float lambda_h, float lambda_g, float threshold, int64_t panel_rows,
                                 int64_t* edge_index, float* edge_w, int64_t capacity, int64_t* out_count,
                                 int device_id, void* hip_stream);

/*
 * Cluster-shaped steps around the similarity kernels: what the reference does with a vector
 * of KMeans labels in Python loops.  labels / order / offsets / counts / pairs are int64 device arrays (torch.long);
 * at most 16384 segments.
 *   mmf_segment_sort          labels[n] in [0, S) -> counts[S], offsets[S+1] (exclusive scan), order[n] = the rows of
 *                             segment 0, then 1, ... each in ascending row order (stable counting sort).  A label
 *                             outside [0, S) -> MMF_E_INVALID.  Host-synchronous (reads that one flag back).
 *   mmf_segment_mean          out[S, d] = mean of the rows of X[n, d] in each segment, summed in member order:
 *                             the per-cluster `features[mask].mean(0)` loop, preprocess_hypergraph.py:157-170.
 *   mmf_segment_offdiag_mean  out_mean[S] (double) = mean of K[i][j] over the ordered pairs i != j of a segment's
 *                             members (NaN for segments with fewer than two): the K[idx][:, idx] off-diagonal means
 *                             of preprocess_hypergraph.py:175-184, by direct gathers (no cancellation).
 *   mmf_clique_pairs          every pair (a < b) inside every segment — the m(m-1) ordered pairs of :395-400 after the
 *                             undirected dedup of :403 — cluster by cluster; writes at most `capacity` pairs, always
 *                             the true count to *out_count (capacity 0 = count only).
 *   mmf_knn_pairs             nbr[n, k] (neighbour ids, self already excluded) -> the undirected pairs (min, max) of
 *                             :386-388 without duplicates: a pair is dropped when both rows emit it (kept from the
 *                             smaller row) or when `labels` (may be NULL) puts both ends in one segment, i.e. a clique
 *                             already holds it.  pair_lo / pair_hi need room for n * k; order unspecified.
 */
/*
 * The reference's clustering call, KMeans(n_clusters, random_state=seed, n_init=n_init).fit_predict(X)
 * (build_hypergraph/preprocess_hypergraph.py:150-151, 299-300, 391-392), on the device and DECISION FOR DECISION:
 * scikit-learn's k-means++ seeding, Lloyd iterations, convergence tests and choice of the best restart
 * (sklearn/cluster/_kmeans.py 1.7.2: KMeans.fit, _kmeans_plusplus, _kmeans_single_lloyd), with every inner product and
 * sum taken in float64 on the matrix cores and rounded to float32 exactly where scikit-learn stores a float32.
 * scikit-learn's own float32 sums go through BLAS in a machine-dependent order; wherever one of its decisions is not
 * taken by that rounding noise, this entry takes the same one, so the labels are scikit-learn's
 * (oracle/kmeans_restate.py is the CPU restatement of this contract; tests/golden g5 / g8 hold scikit-learn's labels).
 *
 * The caller supplies scikit-learn's random stream, which is data independent: with rs = numpy.random.RandomState(seed),
 * per restart one rs.random_sample() (the first centre: RandomState.choice(n, p=uniform), passed here as the resulting
 * row index first_centres[n_init]) followed by n_clusters - 1 draws of rs.uniform(size=trials), trials = 2 + int(log(k)):
 * uniforms[n_init][n_clusters - 1][trials].  BOTH ARE HOST ARRAYS.  All restarts advance in lockstep.
 *   X [n, d] f32 device (not modified); labels [n] int64 device; centres [n_clusters, d] f32 device or NULL;
 *   seeds [n_init, n_clusters] int64 device or NULL (the rows chosen as initial centres);
 *   info (host, 7 + 2 n_init doubles, or NULL): best restart, its inertia, its iterations, the absolute tolerance,
 *   draws / trial choices within 4 float32 ulps of going the other way, lockstep iterations, then (inertia, iterations)
 *   per restart.
 * Host-synchronous (one small status read per Lloyd iteration).  n_init * n_clusters <= 16384, trials <= 64.
 */
int mmf_kmeans_fit(const float* X, int64_t n, int64_t d, int64_t n_clusters, int64_t n_init, int trials,
                   const int64_t* first_centres, const double* uniforms, int max_iter, double tol, int64_t* labels,
                   float* centres, int64_t* seeds, double* info, int device_id, void* hip_stream);

int mmf_segment_sort(const int64_t* labels, int64_t n, int64_t n_segments, int64_t* counts, int64_t* offsets,
                     int64_t* order, int device_id, void* hip_stream);
int mmf_segment_mean(const float* X, int64_t n, int64_t d, const int64_t* order, const int64_t* offsets,
                     int64_t n_segments, float* out, int device_id, void* hip_stream);
int mmf_segment_offdiag_mean(const float* K, int64_t n, const int64_t* order, const int64_t* offsets,
                             int64_t n_segments, double* out_mean, int device_id, void* hip_stream);
int mmf_clique_pairs(const int64_t* order, const int64_t* offsets, int64_t n, int64_t n_segments,
                     int64_t* pair_lo, int64_t* pair_hi, int64_t capacity, int64_t* out_count,
                     int device_id, void* hip_stream);
int mmf_knn_pairs(const int64_t* nbr, int64_t n, int k, const int64_t* labels,
                  int64_t* pair_lo, int64_t* pair_hi, int64_t* out_count, int device_id, void* hip_stream);

/* Release the library's cached per-device workspaces (they are grow-only otherwise). */
int mmf_release_workspaces(void);

#ifdef __cplusplus
}
#endif
#endif /* MMF_HG_H */
