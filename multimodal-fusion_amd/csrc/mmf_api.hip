// mmf_api.hip — the C ABI of include/mmf_hg.h.  Host code only: validation, workspace layout,
// kernel sequencing.  No CPU compute path exists here (device_id < 0 is an error).
#include <stdarg.h>
#include <string.h>

#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "mmf_host.h"

namespace mmf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

struct WsEntry { char* base = nullptr; size_t cap = 0; };
static std::mutex g_ws_mu;
static std::map<std::pair<int, hipStream_t>, WsEntry> g_ws[2];

// slot 0: the call's working set.  slot 1: what only a rare branch of a call needs on top of it (the f32 operand
// images of an exact rescan inside the 16-bit path), so that every call does not carry it.
static int get_workspace_slot(int device, hipStream_t stream, int slot, size_t bytes, Workspace* out) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  WsEntry& e = g_ws[slot][std::make_pair(device, stream)];
  if (e.cap < bytes) {
    if (e.base) {
      MMF_HIP(hipStreamSynchronize(stream));
      MMF_HIP(hipFree(e.base));
      e.base = nullptr;
      e.cap = 0;
    }
    size_t want = bytes + (bytes >> 3) + (1u << 20);
    void* p = nullptr;
    hipError_t rc = hipMalloc(&p, want);
    if (rc != hipSuccess) {
      set_error("workspace allocation of %zu bytes failed: %s", want, hipGetErrorString(rc));
      return MMF_E_NOMEM;
    }
    e.base = static_cast<char*>(p);
    e.cap = want;
  }
  out->base = e.base;
  out->cap = e.cap;
  out->off = 0;
  return MMF_OK;
}
int get_workspace(int device, hipStream_t stream, size_t bytes, Workspace* out) { return get_workspace_slot(device, stream, 0, bytes, out); }

int launch_edge_cosine_impl(const void* X, int64_t d, int dtype, const float* nrm, const int64_t* ei, int64_t E,
                            float* out, hipStream_t s);

// mmf_scan_bf16.hip (fast path)
int scan_bf16_supported(int64_t d, int kk, int dtype);
int scan_bf16_cap(int kk, int dp);
int scan_bf16_dp(int64_t d);
int launch_prep_half(const void* X, int64_t n, int64_t d, int dtype, int metric, const float* scal,
                     const uint32_t* max_n, void* Z,
                     int64_t n_pad, int dp, int z_f16, float* zn, float* rn, float* un, float* cb, uint32_t* maxima,
                     hipStream_t s);
int launch_scan_b16(const void* ZQ, const void* ZC, const float* cb, const float* q_zn, const float* q_rn,
                    const float* q_un, const uint32_t* maxima, int64_t n_rows, int64_t m, int64_t m_pad, int dp,
                    int64_t d, bool f16, int metric, int kk, int col_splits, const CandLists& L, void* scratch,
                    const ScanB16Panel& pn, hipStream_t s, int* grid_out);
int launch_scan_b16_audit(const ScanB16Panel& pn, uint32_t* overflow, int64_t n_rows, hipStream_t s);
size_t scan_b16_scratch_bytes(int64_t n_rows, int col_splits, int dp, int cap);
int scan_bf16_slot_ulp(int cap);
int scan_b16_queries_per_block(int dp);

static int check_common(const void* X, int64_t n, int64_t m, int64_t d, int in_dtype, int device_id) {
  if (device_id < 0) {
    set_error("device_id %d: this library has no CPU path (the CPU restatement is oracle/, tests only)", device_id);
    return MMF_E_UNSUPPORTED;
  }
  if (n < 0 || m < 0 || d < 1) { set_error("bad shape n=%lld m=%lld d=%lld", (long long)n, (long long)m, (long long)d); return MMF_E_INVALID; }
  if (in_dtype != MMF_F32 && in_dtype != MMF_BF16 && in_dtype != MMF_F16) { set_error("bad in_dtype %d", in_dtype); return MMF_E_INVALID; }
  if (n > 0 && !X) { set_error("X is NULL"); return MMF_E_INVALID; }
  if (n >= (int64_t)1 << 31 || m >= (int64_t)1 << 31) { set_error("n and m must be < 2^31"); return MMF_E_UNSUPPORTED; }
  return MMF_OK;
}

// HIP events are kept between calls (per device): creating and destroying eight of them per step showed up next to
// an 8 ms per-rank step.  An EventTimer borrows two from the pool and hands them back when it goes out of scope.
struct EventPool {
  std::mutex mu;
  std::map<int, std::vector<hipEvent_t>> idle;
  hipEvent_t get(int dev) {
    {
      std::lock_guard<std::mutex> lk(mu);
      auto& v = idle[dev];
      if (!v.empty()) { hipEvent_t e = v.back(); v.pop_back(); return e; }
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
  }
  void put(int dev, hipEvent_t e) {
    if (!e) return;
    std::lock_guard<std::mutex> lk(mu);
    idle[dev].push_back(e);
  }
};
static EventPool g_events;

struct EventTimer {
  hipEvent_t a = nullptr, b = nullptr;
  bool on = false;
  int dev = 0;
  int start(bool enable, hipStream_t s) {
    on = enable;
    if (!on) return MMF_OK;
    MMF_HIP(hipGetDevice(&dev));
    a = g_events.get(dev);
    b = g_events.get(dev);
    if (!a || !b) { set_error("hipEventCreate failed"); return MMF_E_HIP; }
    MMF_HIP(hipEventRecord(a, s));
    return MMF_OK;
  }
  int stop(hipStream_t s) {
    if (!on) return MMF_OK;
    MMF_HIP(hipEventRecord(b, s));
    return MMF_OK;
  }
  float ms() {
    float t = 0.f;
    if (on && a && b && hipEventSynchronize(b) == hipSuccess) (void)hipEventElapsedTime(&t, a, b);
    return t;
  }
  ~EventTimer() {
    g_events.put(dev, a);
    g_events.put(dev, b);
  }
};

static int pick_splits(int64_t row_blocks, int64_t col_tiles, int lists_cap, int cap, int forced) {
  int64_t s = forced > 0 ? forced : (768 + row_blocks - 1) / row_blocks;
  if (s > col_tiles) s = col_tiles;
  const int64_t max_lists = 1024 / cap;  // select kernel capacity per row
  if (2 * s > max_lists) s = max_lists / 2;
  if (s < 1) s = 1;
  (void)lists_cap;
  return (int)s;
}

// Everything of the fast path after the 16-bit operands exist: candidate lists, scan, (optional event
// wait), exact re-rank, exact rescan of flagged rows, stats.  Shared by mmf_simtopk_ex and
// mmf_simtopk_prepared.
struct FastOperands {
  const uint16_t* ZQ; const uint16_t* ZC;
  const float* rx; const float* cy;
  const float* q_zn; const float* q_rn; const float* q_un;
  const float* c_cb; const uint32_t* max_c;
  int64_t m_pad_tiles;   // candidate rows covered by tiles (multiple of 256)
  int dp; bool f16;
  // paneled scan (mmf_simtopk_panels): the candidate operands come as n_panels separate blocks, each scanned by
  // its own launch once its event has fired; ZC / c_cb / m_pad_tiles above are then unused
  const mmf_panel* panels = nullptr; int n_panels = 0;
  // ZQ / q_zn / q_rn / q_un hold the queries in scan order: position p is row perm[p] (mmf_order.hip); null: row order
  const int32_t* perm = nullptr;
};

struct FastTail {
  int64_t n, m; int kk, cap, bcap, splits, lists, fb_splits, fb_lists; int64_t FB;
  CandLists L, FL;
  int32_t *fail_rows = nullptr, *fb_fail_rows = nullptr;
  uint32_t *fail_count = nullptr, *cand_total = nullptr, *fb_fail_count = nullptr;
  char* scan_scratch = nullptr;
  char* order_scratch = nullptr;
  // query order of the scan (mmf_order.hip): near-duplicate rows next to each other.  Tried when forced, or (auto) at sizes where
  // the scan dominates; applied when the rows have near-duplicates among themselves (decided from the data, one host sync).
  int order_mode = MMF_QUERY_ORDER_OFF; bool order_try = false;
  char* qo_scratch = nullptr; uint16_t* qo_Z = nullptr; float *qo_zn = nullptr, *qo_rn = nullptr, *qo_un = nullptr;
  int64_t n_pad_q() const { return (n + 255) / 256 * 256; }
  int set_query_order(int mode) {   // before bytes() / carve()
    if (mode < MMF_QUERY_ORDER_AUTO || mode > MMF_QUERY_ORDER_ON) { set_error("simtopk: bad query_order %d", mode); return MMF_E_INVALID; }
    order_mode = mode;
    order_try = mode == MMF_QUERY_ORDER_ON || (mode == MMF_QUERY_ORDER_AUTO && n >= 32768 && m >= 32768);
    return MMF_OK;
  }

  int dp, panels;
  int panel_splits[16]; int max_splits = 1;
  int64_t n_seed;
  int32_t* seed = nullptr;
  // a handful of flagged rows skips the matrix-core rescan: all their keys, then a block-wide top-k
  static constexpr int64_t kRowsExactMax = 48;
  // overflow slots per row for the columns a row's lane lists cannot hold (near-duplicate data); a row that fills
  // them as well is redone exactly
  static constexpr int kSpillCap = 192;
  int64_t rows_exact_cap = 0;
  float* row_keys = nullptr;
  // m_panel_min: columns of the smallest panel (== m_ when the scan is one launch); `splits` is per launch
  FastTail(int64_t n_, int64_t m_, int kk_, int cap_, int forced_splits, int dp_, int panels_ = 1, int64_t m_panel_min = -1,
           int64_t m_panel_max = -1)
      : n(n_), m(m_), kk(kk_), cap(cap_), dp(dp_), panels(panels_) {
    bcap = scan_bf16_cap(kk, dp);
    const int qt = scan_b16_queries_per_block(dp);
    if (m_panel_min < 0) m_panel_min = m;
    const int64_t row_blocks = (n + qt - 1) / qt, col_tiles = ((m_panel_min + 255) / 256 * 256) / 32;
    n_seed = row_blocks * qt;
    splits = 1;
    if (forced_splits > 0) { while (splits < forced_splits) splits <<= 1; }
    else { while (row_blocks * splits < 256 && splits < 32) splits <<= 1; }
    while (splits > 1 && (splits > col_tiles || 2 * splits * panels * bcap > 1024 - kSpillCap)) splits >>= 1;
    // the tile DMA addresses a workgroup's column range with 32-bit offsets: a range stays under 4 GiB of 16-bit operands
    // (N = 4 M rows at d = 1024 is 8 GiB: at least four splits), whatever the caller forced
    {
      const int64_t m_big = (m_panel_max > 0 ? m_panel_max : m);
      const int64_t range_bytes = ((m_big + 255) / 256 * 256) * (int64_t)dp * 2;
      while ((range_bytes + splits - 1) / splits >= (int64_t(1) << 32) && 2 * (2 * splits) * panels * bcap <= 1024 - kSpillCap) splits <<= 1;
    }
    // The first half of the panels runs while the rest of the exchange is still on the wire and its kernel
    // holds compute units.  MMF_PANEL_FRONT_FACTOR = 2 or 4 gives those launches that many times the workgroups
    // (shorter ones), which shortens the tail the late-joining units leave — measured with a stand-in kernel
    // (scripts/overlap_sim.py) it trades 0.6 ms without contention for 0.9 ms with it, so the default stays 1.
    int front = 1;
    if (const char* e = getenv("MMF_PANEL_FRONT_FACTOR")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) front = v; }
    int total_splits = 0;
    for (int p = 0; p < panels && p < 16; ++p) {
      int sp = splits * ((panels > 1 && p < panels / 2) ? front : 1);
      while (sp > 1 && sp > col_tiles) sp >>= 1;
      panel_splits[p] = sp;
      total_splits += sp;
    }
    if (2 * total_splits * bcap > 1024 - kSpillCap) { // too many lists for the select kernel: uniform
      total_splits = splits * panels;
      for (int p = 0; p < panels && p < 16; ++p) panel_splits[p] = splits;
    }
    max_splits = 1;
    for (int p = 0; p < panels && p < 16; ++p) if (panel_splits[p] > max_splits) max_splits = panel_splits[p];
    lists = 2 * total_splits;
    rows_exact_cap = (int64_t(64) << 20) / (4 * (m > 0 ? m : 1));
    if (rows_exact_cap > kRowsExactMax) rows_exact_cap = kRowsExactMax;
    if (rows_exact_cap < 1) rows_exact_cap = 1;
    FB = n < 4096 ? n : 4096;   // exact rescans are done in batches of at most FB rows
    fb_splits = pick_splits((FB + 127) / 128, (m + 127) / 128, 0, cap, 0);
    fb_lists = 2 * fb_splits;
  }
  size_t bytes() const {
    return ws_bytes((size_t)n * lists, 4) + 2 * ws_bytes((size_t)n * lists * bcap, 4) + 3 * ws_bytes(n, 4) + ws_bytes(4, 4) +
           ws_bytes(256, 4) + ws_bytes((size_t)FB * fb_lists, 4) + ws_bytes((size_t)FB * fb_lists * cap, 4) +
           2 * ws_bytes(FB, 4) + ws_bytes(4, 4) + ws_bytes(scan_b16_scratch_bytes(n, max_splits, dp, bcap), 1) + ws_bytes(2 * n_seed, 4) +
           ws_bytes((size_t)rows_exact_cap * m, 4) + ws_bytes(n, 4) + ws_bytes((size_t)n * kSpillCap, 4) + ws_bytes(select_order_bytes(n), 1) +
           (order_try ? ws_bytes(query_order_bytes(n), 1) + ws_bytes((size_t)n_pad_q() * dp, 2) + 3 * ws_bytes(n_pad_q(), 4) : 0);
  }
  void carve(Workspace& ws) {
    L.cnt = ws.take<uint32_t>((size_t)n * lists);
    L.ids = ws.take<uint32_t>((size_t)n * lists * bcap);
    L.keys = ws.take<float>((size_t)n * lists * bcap);
    L.margin = ws.take<float>(n);
    L.overflow = ws.take<uint32_t>(n);
    L.lists = lists; L.cap = bcap; L.slot_ulp = scan_bf16_slot_ulp(bcap);
    fail_rows = ws.take<int32_t>(n);
    fail_count = ws.take<uint32_t>(4);
    cand_total = ws.take<uint32_t>(256);
    FL.cnt = ws.take<uint32_t>((size_t)FB * fb_lists);
    FL.ids = ws.take<uint32_t>((size_t)FB * fb_lists * cap);
    FL.overflow = ws.take<uint32_t>(FB);
    FL.lists = fb_lists; FL.cap = cap;
    fb_fail_rows = ws.take<int32_t>(FB);
    fb_fail_count = ws.take<uint32_t>(4);
    scan_scratch = ws.take<char>(scan_b16_scratch_bytes(n, max_splits, dp, bcap));
    seed = ws.take<int32_t>(2 * n_seed);
    if (lists <= 2) { L.keys = nullptr; L.margin = nullptr; }    // one list pair per row: nothing to prune against
    row_keys = ws.take<float>((size_t)rows_exact_cap * m);
    L.spill_cnt = ws.take<uint32_t>(n);
    L.spill_ids = ws.take<uint32_t>((size_t)n * kSpillCap);
    L.spill_cap = kSpillCap;
    L.spill_stacks = (lists == 2 && kSpillCap < 65536 && !getenv("MMF_SPILL_COUNTER")) ? 1 : 0;
    order_scratch = ws.take<char>(select_order_bytes(n));
    if (order_try) {
      qo_scratch = ws.take<char>(query_order_bytes(n));
      qo_Z = ws.take<uint16_t>((size_t)n_pad_q() * dp);
      qo_zn = ws.take<float>(n_pad_q()); qo_rn = ws.take<float>(n_pad_q()); qo_un = ws.take<float>(n_pad_q());
    }
  }
  int run(const void* X, int64_t n_, const void* Y, int64_t m_, int64_t d, int in_dtype, int metric, float lambda, int k,
          int exclude_self, int64_t row_offset, int64_t col_offset, const FastOperands& fo_in, int64_t* out_idx,
          float* out_val, bool profile, void* select_wait_event, mmf_simtopk_stats* stats, int precision, hipStream_t s) {
    FastOperands fo = fo_in;
    query_order_forget();
    EventTimer t_order;
    int64_t near_rows = -1;
    if (order_try) {
      MMF_TRY(t_order.start(profile, s));
      MMF_TRY(launch_query_order_probe(fo.ZQ, fo.q_zn, n, fo.dp, fo.f16, qo_scratch, &near_rows, s));
      if (order_mode == MMF_QUERY_ORDER_ON || near_rows >= 8 * (int64_t)query_order_pivots()) {
        MMF_TRY(launch_query_order_apply(fo.ZQ, fo.q_zn, fo.q_rn, fo.q_un, n, n_pad_q(), fo.dp, fo.f16, qo_scratch, qo_Z, qo_zn, qo_rn, qo_un, &fo.perm, s));
        fo.ZQ = qo_Z; fo.q_zn = qo_zn; fo.q_rn = qo_rn; fo.q_un = qo_un;
      }
      MMF_TRY(t_order.stop(s));
    }
    MMF_HIP(hipMemsetAsync(L.overflow, 0, (size_t)n * 4, s));
    MMF_HIP(hipMemsetAsync(fail_count, 0, 16, s));
    MMF_HIP(hipMemsetAsync(cand_total, 0, 1024, s));
    EventTimer t_scan, t_sel, t_fb;
    EventTimer t_panel[16];
    int grid = 0;
    MMF_HIP(hipMemsetAsync(seed, 0x80, (size_t)n_seed * 8, s));   // kSeedNone, thresholds and dropped keys
    MMF_HIP(hipMemsetAsync(L.spill_cnt, 0, (size_t)n * 4, s));
    MMF_TRY(t_scan.start(profile, s));
    ScanB16Panel pn;
    pn.seed = seed; pn.seed_stride = n_seed;
    pn.share = (splits > 1 || fo.n_panels > 1) ? 1 : 0;
    if (fo.n_panels == 0) {
      MMF_TRY(launch_scan_b16(fo.ZQ, fo.ZC, fo.c_cb, fo.q_zn, fo.q_rn, fo.q_un, fo.max_c, n, m, fo.m_pad_tiles, fo.dp, d, fo.f16,
                              metric, kk, splits, L, scan_scratch, pn, s, &grid));
    } else {
      // one launch per panel, each behind its own arrival event; the launches share the lists (disjoint
      // slots), the id scratch (they run one after the other) and the per-query thresholds
      int list_base = 0;
      for (int p = 0; p < fo.n_panels; ++p) {
        const mmf_panel& P = fo.panels[p];
        if (P.ready_event) MMF_HIP(hipStreamWaitEvent(s, static_cast<hipEvent_t>(P.ready_event), 0));
        MMF_TRY(t_panel[p].start(profile, s));       // behind the wait: launch-only time of this panel
        pn.list_base = list_base;
        list_base += 2 * panel_splits[p];
        pn.seg_len = (uint32_t)P.seg_len; pn.seg_stride = (uint32_t)P.seg_stride; pn.id_off = (uint32_t)P.id_base;
        MMF_TRY(launch_scan_b16(fo.ZQ, P.Z, P.cb, fo.q_zn, fo.q_rn, fo.q_un, fo.max_c, n, P.m, P.m_pad, fo.dp, d, fo.f16,
                                metric, kk, panel_splits[p], L, scan_scratch, pn, s, &grid));
        MMF_TRY(t_panel[p].stop(s));
      }
    }
    MMF_TRY(launch_scan_b16_audit(pn, L.overflow, n, s));
    if (const char* e = getenv("MMF_DEBUG_FLAG_ROWS")) {   // test hook: send the first rows down the exact paths
      const int64_t f = atoll(e);
      if (f > 0) MMF_HIP(hipMemsetAsync(L.overflow, 1, (size_t)(f < n ? f : n) * 4, s));
    }
    MMF_TRY(t_scan.stop(s));
    // the f32 rows are first touched here: a caller that is still receiving them (overlapped
    // all-gather) hands in the event that marks their arrival
    if (select_wait_event) MMF_HIP(hipStreamWaitEvent(s, static_cast<hipEvent_t>(select_wait_event), 0));

    SelectProblem q{};
    q.X = X; q.n = n; q.Y = Y; q.m = m; q.d = d; q.dtype = in_dtype; q.metric = metric; q.lambda = lambda;
    q.k = k; q.exclude_self = exclude_self; q.row_offset = row_offset; q.col_offset = col_offset;
    q.rx = fo.rx; q.cy = fo.cy; q.row_ids = nullptr; q.perm = fo.perm; q.n_rows = n; q.out_idx = out_idx; q.out_val = out_val;
    q.fail_rows = fail_rows; q.fail_count = fail_count; q.cand_total = stats ? cand_total : nullptr;
    q.two_pass = true;
    q.order_scratch = getenv("MMF_SELECT_UNORDERED") ? nullptr : order_scratch;
    MMF_TRY(t_sel.start(profile, s));
    MMF_TRY(launch_select(q, L, s));
    MMF_TRY(t_sel.stop(s));

    uint32_t h_fail4[4] = {0, 0, 0, 0};
    MMF_HIP(hipMemcpyAsync(h_fail4, fail_count, 16, hipMemcpyDeviceToHost, s));
    std::vector<uint32_t> h_tot(stats ? 256 : 0);
    if (stats) MMF_HIP(hipMemcpyAsync(h_tot.data(), cand_total, 1024, hipMemcpyDeviceToHost, s));
    MMF_HIP(hipStreamSynchronize(s));
    const uint32_t h_fail = h_fail4[0];
    if (h_fail > 0 && getenv("MMF_DEBUG_PRINT_FLAGGED")) {   // diagnosis: which rows, and the threshold / dropped key that flagged them
      const uint32_t nshow = h_fail < 8 ? h_fail : 8;
      int32_t rows[8];
      MMF_HIP(hipMemcpy(rows, fail_rows, nshow * sizeof(int32_t), hipMemcpyDeviceToHost));
      for (uint32_t i = 0; i < nshow; ++i) {
        int32_t enc[2] = {0, 0};
        MMF_HIP(hipMemcpy(&enc[0], seed + rows[i], 4, hipMemcpyDeviceToHost));
        MMF_HIP(hipMemcpy(&enc[1], seed + n_seed + rows[i], 4, hipMemcpyDeviceToHost));
        auto dec = [](int32_t o) { int32_t b = o >= 0 ? o : (o ^ 0x7fffffff); float f; memcpy(&f, &b, 4); return f; };
        fprintf(stderr, "[mmf flagged] row %d: best threshold %.9g (enc %d)  best dropped key %.9g (enc %d)  overflow %u short %u\n",
                rows[i], dec(enc[0]), enc[0], dec(enc[1]), enc[1], h_fail4[1], h_fail4[2]);
      }
    }

    MMF_TRY(t_fb.start(profile && h_fail > 0, s));
    if (h_fail > 0 && (int64_t)h_fail <= kRowsExactMax) {
      for (int64_t off = 0; off < (int64_t)h_fail; off += rows_exact_cap) {
        SelectProblem fq = q;
        fq.perm = nullptr;
        fq.row_ids = fail_rows + off;
        fq.n_rows = ((int64_t)h_fail - off < rows_exact_cap) ? ((int64_t)h_fail - off) : rows_exact_cap;
        MMF_TRY(launch_rows_exact(fq, row_keys, s));
      }
    } else {
    float *fb_Xp = nullptr, *fb_Yp = nullptr;
    int dev_now = 0;
    MMF_HIP(hipGetDevice(&dev_now));
    for (int64_t off = 0; off < (int64_t)h_fail; off += FB) {
      const int64_t nb = ((int64_t)h_fail - off < FB) ? ((int64_t)h_fail - off) : FB;
      if (off == 0) {   // f32 operand images for the exact scan: all candidate rows once, the flagged rows per batch
        Workspace aux;
        MMF_TRY(get_workspace_slot(dev_now, s, 1, ws_bytes(prep_f32_bytes(m, d), 1) + ws_bytes(prep_f32_bytes(FB, d), 1), &aux));
        fb_Yp = reinterpret_cast<float*>(aux.take<char>(prep_f32_bytes(m, d)));
        fb_Xp = reinterpret_cast<float*>(aux.take<char>(prep_f32_bytes(FB, d)));
        MMF_TRY(launch_prep_f32(Y, m, d, in_dtype, nullptr, fb_Yp, s));
      }
      MMF_TRY(launch_prep_f32(X, nb, d, in_dtype, fail_rows + off, fb_Xp, s));
      MMF_HIP(hipMemsetAsync(FL.overflow, 0, (size_t)nb * 4, s));
      MMF_HIP(hipMemsetAsync(fb_fail_count, 0, 16, s));
      ScanProblem sp{};
      sp.X = X; sp.n = n; sp.Y = Y; sp.m = m; sp.d = d; sp.dtype = in_dtype; sp.metric = metric; sp.lambda = lambda;
      sp.Xp = fb_Xp; sp.Yp = fb_Yp;
      sp.kk = kk; sp.rx = fo.rx; sp.cy = fo.cy; sp.row_ids = fail_rows + off; sp.n_rows = nb; sp.col_splits = fb_splits;
      MMF_TRY(launch_scan_f32(sp, FL, s, nullptr));
      SelectProblem fq = q;
      fq.perm = nullptr;
      fq.row_ids = fail_rows + off; fq.n_rows = nb; fq.fail_rows = fb_fail_rows; fq.fail_count = fb_fail_count;
      fq.cand_total = nullptr;
      MMF_TRY(launch_select(fq, FL, s));
      uint32_t h_fb = 0;
      MMF_HIP(hipMemcpyAsync(&h_fb, fb_fail_count, 4, hipMemcpyDeviceToHost, s));
      MMF_HIP(hipStreamSynchronize(s));
      if (h_fb != 0) {
        set_error("simtopk: %u rows failed in the exact rescan (internal invariant)", h_fb);
        return MMF_E_INTERNAL;
      }
    }
    }
    MMF_TRY(t_fb.stop(s));
    if (stats) {
      stats->precision_used = precision;
      stats->col_splits = splits;
      stats->scan_grid = grid;
      stats->scan_ms = t_scan.ms();
      stats->rerank_ms = t_sel.ms();
      stats->fallback_ms = t_fb.ms();
      if (profile && fo.n_panels > 0) {     // what the scan stream spent waiting for panels to arrive
        float launches = 0.f;
        for (int p = 0; p < fo.n_panels; ++p) launches += t_panel[p].ms();
        const float w = stats->scan_ms - launches;
        stats->scan_wait_ms = w > 0.f ? w : 0.f;
      }
      stats->fallback_rows = h_fail;
      stats->overflow_rows = h_fail4[1];
      stats->short_rows = h_fail4[2];
      int64_t tot = 0;
      for (uint32_t v : h_tot) tot += v;
      stats->candidates = tot;
      stats->near_rows = near_rows;
      stats->order_ms = t_order.ms();
      stats->query_order = fo.perm ? 1 : 0;
    }
    (void)n_; (void)m_;
    return MMF_OK;
  }
};

}  // namespace mmf

using namespace mmf;

extern "C" {

int mmf_version(void) { return MMF_ABI_VERSION; }
int mmf_debug_query_order(int32_t* perm_host, int64_t n) { return query_order_last(perm_host, n); }
const char* mmf_last_error(void) { return g_err; }

int mmf_release_workspaces(void) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  query_order_forget();
  for (auto& slot : g_ws) {
    for (auto& kv : slot) {
      if (kv.second.base) {
        int prev = -1;
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(kv.first.first);
        (void)hipDeviceSynchronize();
        (void)hipFree(kv.second.base);
        if (prev >= 0) (void)hipSetDevice(prev);
      }
    }
    slot.clear();
  }
  return MMF_OK;
}

int mmf_simtopk_ex(const void* X, int64_t n, const void* Y, int64_t m, int64_t d, int in_dtype, int metric,
                   float lambda, int k, int exclude_self, int64_t row_offset, int64_t col_offset, int64_t* out_idx,
                   float* out_val, const mmf_simtopk_opts* opts, mmf_simtopk_stats* stats, int device_id,
                   void* hip_stream) {
  if (!Y) { Y = X; m = n; }
  MMF_TRY(check_common(X, n, m, d, in_dtype, device_id));
  if (metric < MMF_DOT || metric > MMF_RBF) { set_error("simtopk: bad metric %d", metric); return MMF_E_INVALID; }
  if (metric == MMF_RBF && !(lambda > 0.0f)) { set_error("simtopk: MMF_RBF needs lambda > 0 (got %g)", lambda); return MMF_E_INVALID; }
  if (k < 1) { set_error("simtopk: k must be >= 1 (got %d)", k); return MMF_E_INVALID; }
  if (stats) memset(stats, 0, sizeof(*stats));
  if (n == 0) return MMF_OK;
  if (!out_idx || !out_val) { set_error("simtopk: NULL output"); return MMF_E_INVALID; }
  // admissible columns: m, minus one for rows whose own id lies in the column range
  {
    const int64_t lo = col_offset, hi = col_offset + m;
    const int64_t r0 = row_offset, r1 = row_offset + n;  // any overlap -> some row loses one column
    const bool overlap = exclude_self && (r0 < hi) && (r1 > lo);
    const int64_t adm = m - (overlap ? 1 : 0);
    if (k > adm) {
      set_error("simtopk: k = %d exceeds the %lld admissible columns (m = %lld%s)", k, (long long)adm, (long long)m,
                overlap ? ", self excluded" : "");
      return MMF_E_INVALID;
    }
  }
  const int kk = k + (exclude_self ? 1 : 0);
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }

  int precision = opts ? opts->precision : MMF_PREC_AUTO;
  const bool profile = opts && opts->profile;
  const int forced_splits = opts ? opts->col_splits : 0;
  if (precision == MMF_PREC_AUTO) precision = scan_bf16_supported(d, kk, in_dtype) ? MMF_PREC_FAST : MMF_PREC_EXACT;
  if (precision != MMF_PREC_EXACT && precision != MMF_PREC_FAST && precision != MMF_PREC_FAST_BF16) {
    set_error("simtopk: bad precision %d", precision);
    return MMF_E_INVALID;
  }
  if (precision != MMF_PREC_EXACT && !scan_bf16_supported(d, kk, in_dtype)) {
    set_error("simtopk: MMF_PREC_FAST does not support d = %lld, k = %d (use AUTO or EXACT)", (long long)d, k);
    return MMF_E_UNSUPPORTED;
  }
  // k + self beyond 44 (one pass of the exact scan): several passes, each offering only what ranks after the previous pass's
  // last entry (scikit-learn's n_neighbors is uncapped, preprocess_hypergraph.py:379)
  const int kk_pass = kk <= 44 ? kk : 44;
  const int cap = scan_f32_cap(kk_pass);
  if (cap == 0) { set_error("simtopk: no list capacity for k = %d (internal)", k); return MMF_E_INTERNAL; }

  if (precision != MMF_PREC_EXACT) {
    // ---- fast path: f16/bf16 MFMA scan -> exact re-rank -> exact rescan of overflowed rows -------
    const int dp = scan_bf16_dp(d);
    const bool f16 = (precision == MMF_PREC_FAST);   // operand type of the scan, not of the input
    const bool same = (Y == X) && (m == n);
    // X a row-slice of Y (the row-sharded multi-GPU case passes full[lo:hi] and full): every query-side
    // buffer is then a view into the candidate-side one and only Y is prepared.
    const size_t row_bytes = (size_t)d * dtype_size(in_dtype);
    int64_t slice0 = -1;
    if (!same) {
      const char* xb = static_cast<const char*>(X);
      const char* yb = static_cast<const char*>(Y);
      if (xb >= yb && xb + (size_t)n * row_bytes <= yb + (size_t)m * row_bytes && ((size_t)(xb - yb) % row_bytes) == 0)
        slice0 = (int64_t)((size_t)(xb - yb) / row_bytes);
    }
    const bool shared = same || slice0 >= 0;
    const int64_t n_pad = (n + 255) / 256 * 256, m_pad = (m + 255) / 256 * 256 + (slice0 >= 0 ? 256 : 0);
    FastTail ft(n, m, kk, cap, forced_splits, dp);
    MMF_TRY(ft.set_query_order(opts ? opts->query_order : MMF_QUERY_ORDER_AUTO));
    size_t need = ws_bytes(n, 4) + ws_bytes(m, 4) + ws_bytes((size_t)n_pad * dp, 2) + ws_bytes((size_t)m_pad * dp, 2) +
                  4 * ws_bytes(n_pad, 4) + 4 * ws_bytes(m_pad, 4) + 3 * ws_bytes(4, 4) + ft.bytes();
    Workspace ws;
    MMF_TRY(get_workspace(device_id, s, need, &ws));
    float *rx, *cy, *q_zn, *q_rn, *q_un, *q_cb, *c_zn, *c_rn, *c_un, *c_cb;
    uint16_t *ZQ, *ZC;
    uint32_t *max_q, *max_c;
    if (shared) {
      const int64_t r0 = same ? 0 : slice0;
      cy = ws.take<float>(m); rx = cy + r0;
      ZC = ws.take<uint16_t>((size_t)m_pad * dp); ZQ = ZC + (size_t)r0 * dp;
      c_zn = ws.take<float>(m_pad); c_rn = ws.take<float>(m_pad); c_un = ws.take<float>(m_pad); c_cb = ws.take<float>(m_pad);
      q_zn = c_zn + r0; q_rn = c_rn + r0; q_un = c_un + r0; q_cb = c_cb + r0;
      max_c = ws.take<uint32_t>(4); max_q = max_c;
    } else {
      rx = ws.take<float>(n); cy = ws.take<float>(m);
      ZQ = ws.take<uint16_t>((size_t)n_pad * dp); ZC = ws.take<uint16_t>((size_t)m_pad * dp);
      q_zn = ws.take<float>(n_pad); q_rn = ws.take<float>(n_pad); q_un = ws.take<float>(n_pad); q_cb = ws.take<float>(n_pad);
      c_zn = ws.take<float>(m_pad); c_rn = ws.take<float>(m_pad); c_un = ws.take<float>(m_pad); c_cb = ws.take<float>(m_pad);
      max_q = ws.take<uint32_t>(4); max_c = ws.take<uint32_t>(4);
    }
    uint32_t* max_n = ws.take<uint32_t>(4);   // largest squared row norm over X and Y -> common scale
    ft.carve(ws);
    MMF_HIP(hipMemsetAsync(max_c, 0, 16, s));
    MMF_HIP(hipMemsetAsync(max_n, 0, 16, s));
    if (!shared) MMF_HIP(hipMemsetAsync(max_q, 0, 16, s));

    EventTimer t_prep;
    MMF_TRY(t_prep.start(profile, s));
    MMF_TRY(launch_row_scalars(Y, m, d, in_dtype, metric, cy, max_n, s));
    if (!shared) MMF_TRY(launch_row_scalars(X, n, d, in_dtype, metric, rx, max_n, s));
    MMF_TRY(launch_prep_half(Y, m, d, in_dtype, metric, cy, max_n, ZC, m_pad, dp, f16 ? 1 : 0, c_zn, c_rn, c_un, c_cb, max_c, s));
    if (!shared) MMF_TRY(launch_prep_half(X, n, d, in_dtype, metric, rx, max_n, ZQ, n_pad, dp, f16 ? 1 : 0, q_zn, q_rn, q_un, q_cb, max_q, s));
    MMF_TRY(t_prep.stop(s));

    FastOperands fo{ZQ, ZC, rx, cy, q_zn, q_rn, q_un, c_cb, max_c, (m + 255) / 256 * 256, dp, f16};
    MMF_TRY(ft.run(X, n, Y, m, d, in_dtype, metric, lambda, k, exclude_self, row_offset, col_offset, fo, out_idx, out_val,
                   profile, opts ? opts->select_wait_event : nullptr, stats, precision, s));
    if (stats) stats->prep_ms = t_prep.ms();
    return MMF_OK;
  }

  // ---- exact path ------------------------------------------------------------------------------
  const int64_t row_blocks = (n + 127) / 128, col_tiles = (m + 127) / 128;
  const int splits = pick_splits(row_blocks, col_tiles, 0, cap, forced_splits);
  const int lists = 2 * splits;
  const bool same = (Y == X) && (m == n);
  size_t need = ws_bytes(n, 4) + (same ? 0 : ws_bytes(m, 4)) + ws_bytes((size_t)n * lists, 4) +
                ws_bytes((size_t)n * lists * cap, 4) + ws_bytes(n, 4) + ws_bytes(n, 4) + ws_bytes(4, 4) + ws_bytes(256, 4) +
                ws_bytes(prep_f32_bytes(m, d), 1) + (same ? 0 : ws_bytes(prep_f32_bytes(n, d), 1)) + 2 * ws_bytes(n, 4);
  Workspace ws;
  MMF_TRY(get_workspace(device_id, s, need, &ws));
  float* rx = ws.take<float>(n);
  float* cy = same ? rx : ws.take<float>(m);
  CandLists L;
  L.cnt = ws.take<uint32_t>((size_t)n * lists);
  L.ids = ws.take<uint32_t>((size_t)n * lists * cap);
  L.overflow = ws.take<uint32_t>(n);
  L.lists = lists;
  L.cap = cap;
  int32_t* fail_rows = ws.take<int32_t>(n);
  uint32_t* fail_count = ws.take<uint32_t>(4);
  uint32_t* cand_total = ws.take<uint32_t>(256);
  MMF_HIP(hipMemsetAsync(L.overflow, 0, (size_t)n * 4, s));
  MMF_HIP(hipMemsetAsync(fail_count, 0, 16, s));
  MMF_HIP(hipMemsetAsync(cand_total, 0, 1024, s));

  EventTimer t_prep, t_scan, t_sel;
  MMF_TRY(t_prep.start(profile, s));
  MMF_TRY(launch_row_scalars(X, n, d, in_dtype, metric, rx, nullptr, s));
  if (!same) MMF_TRY(launch_row_scalars(Y, m, d, in_dtype, metric, cy, nullptr, s));
  // f32 operand images (mmf_prep.hip): what the exact scan's LDS-DMA copies
  float* Yp = reinterpret_cast<float*>(ws.take<char>(prep_f32_bytes(m, d)));
  float* Xp = same ? Yp : reinterpret_cast<float*>(ws.take<char>(prep_f32_bytes(n, d)));
  MMF_TRY(launch_prep_f32(Y, m, d, in_dtype, nullptr, Yp, s));
  if (!same) MMF_TRY(launch_prep_f32(X, n, d, in_dtype, nullptr, Xp, s));
  MMF_TRY(t_prep.stop(s));

  float* floor_key = ws.take<float>(n);
  uint32_t* floor_id = ws.take<uint32_t>(n);
  const int self1 = exclude_self ? 1 : 0;
  const int k_pass_max = 44 - self1;              // entries one pass can emit
  int grid = 0;
  const bool one_pass = k <= k_pass_max;          // timers: scan and re-rank apart for one pass, the whole loop as "scan" otherwise
  MMF_TRY(t_scan.start(profile, s));
  for (int done = 0; done < k; done += k_pass_max) {
    const int kp = (k - done < k_pass_max) ? (k - done) : k_pass_max;
    const bool more = done + kp < k;
    ScanProblem sp{};
    sp.X = X; sp.n = n; sp.Y = Y; sp.m = m; sp.d = d; sp.dtype = in_dtype; sp.metric = metric; sp.lambda = lambda;
    sp.Xp = Xp; sp.Yp = Yp;
    sp.kk = kp + self1; sp.rx = rx; sp.cy = cy; sp.row_ids = nullptr; sp.n_rows = n; sp.col_splits = splits;
    if (done > 0) { sp.floor_key = floor_key; sp.floor_id = floor_id; }
    MMF_TRY(launch_scan_f32(sp, L, s, &grid));
    if (one_pass) { MMF_TRY(t_scan.stop(s)); MMF_TRY(t_sel.start(profile, s)); }

    SelectProblem q{};
    q.X = X; q.n = n; q.Y = Y; q.m = m; q.d = d; q.dtype = in_dtype; q.metric = metric; q.lambda = lambda;
    q.k = kp; q.exclude_self = exclude_self; q.row_offset = row_offset; q.col_offset = col_offset;
    q.rx = rx; q.cy = cy; q.row_ids = nullptr; q.n_rows = n; q.out_idx = out_idx; q.out_val = out_val;
    q.out_stride = k; q.out_off = done;
    if (more) { q.floor_key_out = floor_key; q.floor_id_out = floor_id; }
    q.fail_rows = fail_rows; q.fail_count = fail_count; q.cand_total = stats ? cand_total : nullptr;
    MMF_TRY(launch_select(q, L, s));
  }
  if (one_pass) { MMF_TRY(t_sel.stop(s)); }
  else { MMF_TRY(t_scan.stop(s)); MMF_TRY(t_sel.start(profile, s)); MMF_TRY(t_sel.stop(s)); }

  uint32_t h_fail = 0;
  MMF_HIP(hipMemcpyAsync(&h_fail, fail_count, 4, hipMemcpyDeviceToHost, s));
  std::vector<uint32_t> h_tot(stats ? 256 : 0);
  if (stats) MMF_HIP(hipMemcpyAsync(h_tot.data(), cand_total, 1024, hipMemcpyDeviceToHost, s));
  MMF_HIP(hipStreamSynchronize(s));
  if (h_fail != 0) {
    set_error("simtopk: %u rows failed in the exact scan (internal invariant)", h_fail);
    return MMF_E_INTERNAL;
  }
  if (stats) {
    stats->precision_used = MMF_PREC_EXACT;
    stats->col_splits = splits;
    stats->scan_grid = grid;
    stats->prep_ms = t_prep.ms();
    stats->scan_ms = t_scan.ms();
    stats->rerank_ms = t_sel.ms();
    int64_t tot = 0;
    for (uint32_t v : h_tot) tot += v;
    stats->candidates = tot;
  }
  return MMF_OK;
}

int mmf_simtopk(const void* X, int64_t n, const void* Y, int64_t m, int64_t d, int in_dtype, int metric, float lambda,
                int k, int exclude_self, int64_t row_offset, int64_t col_offset, int64_t* out_idx, float* out_val,
                int device_id, void* hip_stream) {
  return mmf_simtopk_ex(X, n, Y, m, d, in_dtype, metric, lambda, k, exclude_self, row_offset, col_offset, out_idx,
                        out_val, nullptr, nullptr, device_id, hip_stream);
}

int64_t mmf_padded_dim(int64_t d) { return (int64_t)scan_bf16_dp(d); }

int mmf_fast_scan_supported(int64_t d, int k, int exclude_self) {
  if (d < 1 || k < 1) return 0;
  return scan_bf16_supported(d, k + (exclude_self ? 1 : 0), MMF_F32);
}

int mmf_row_scalars(const void* X, int64_t n, int64_t d, int in_dtype, int metric, float* scal, float* max_sq_norm,
                    int device_id, void* hip_stream) {
  MMF_TRY(check_common(X, n, n, d, in_dtype, device_id));
  if (metric < MMF_DOT || metric > MMF_RBF) { set_error("row_scalars: bad metric %d", metric); return MMF_E_INVALID; }
  if (n == 0) return MMF_OK;
  if (!scal) { set_error("row_scalars: NULL output"); return MMF_E_INVALID; }
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }
  return launch_row_scalars(X, n, d, in_dtype, metric, scal, reinterpret_cast<uint32_t*>(max_sq_norm),
                            static_cast<hipStream_t>(hip_stream));
}

int mmf_prep_rows(const void* X, int64_t n, int64_t d, int in_dtype, int metric, int operand, const float* scal,
                  const float* max_sq_norm, void* Z, int64_t n_pad, float* zn, float* rn, float* un, float* cb,
                  float* maxima, int device_id, void* hip_stream) {
  MMF_TRY(check_common(X, n, n, d, in_dtype, device_id));
  if (metric < MMF_DOT || metric > MMF_RBF) { set_error("prep_rows: bad metric %d", metric); return MMF_E_INVALID; }
  if (operand != MMF_F16 && operand != MMF_BF16) { set_error("prep_rows: operand must be MMF_F16 or MMF_BF16"); return MMF_E_INVALID; }
  const int dp = scan_bf16_dp(d);
  if (dp == 0) { set_error("prep_rows: d = %lld is not supported by the 16-bit scan", (long long)d); return MMF_E_UNSUPPORTED; }
  if (n_pad < n) { set_error("prep_rows: n_pad < n"); return MMF_E_INVALID; }
  if (n_pad == 0) return MMF_OK;
  if (!scal || !Z || !zn || !rn || !un || !cb || !maxima || (metric != MMF_COSINE && !max_sq_norm)) {
    set_error("prep_rows: NULL pointer"); return MMF_E_INVALID;
  }
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }
  return launch_prep_half(X, n, d, in_dtype, metric, scal, reinterpret_cast<const uint32_t*>(max_sq_norm), Z, n_pad, dp,
                          operand == MMF_F16 ? 1 : 0, zn, rn, un, cb, reinterpret_cast<uint32_t*>(maxima),
                          static_cast<hipStream_t>(hip_stream));
}

int mmf_simtopk_prepared(const void* X, int64_t n, const void* Y, int64_t m, int64_t d, int in_dtype, int metric,
                         float lambda, int k, int exclude_self, int64_t row_offset, int64_t col_offset,
                         const mmf_prepared_side* q, const mmf_prepared_side* c, int64_t m_pad, const float* maxima,
                         int operand, int64_t* out_idx, float* out_val, const mmf_simtopk_opts* opts,
                         mmf_simtopk_stats* stats, int device_id, void* hip_stream) {
  MMF_TRY(check_common(X, n, m, d, in_dtype, device_id));
  if (!Y || !q || !c || !maxima) { set_error("simtopk_prepared: NULL pointer"); return MMF_E_INVALID; }
  if (metric < MMF_DOT || metric > MMF_RBF) { set_error("simtopk_prepared: bad metric %d", metric); return MMF_E_INVALID; }
  if (metric == MMF_RBF && !(lambda > 0.0f)) { set_error("simtopk_prepared: MMF_RBF needs lambda > 0"); return MMF_E_INVALID; }
  if (operand != MMF_F16 && operand != MMF_BF16) { set_error("simtopk_prepared: bad operand"); return MMF_E_INVALID; }
  if (k < 1) { set_error("simtopk_prepared: k must be >= 1"); return MMF_E_INVALID; }
  if (m_pad < m || (m_pad % 256) != 0) { set_error("simtopk_prepared: m_pad must be a multiple of 256 and >= m"); return MMF_E_INVALID; }
  if (stats) memset(stats, 0, sizeof(*stats));
  if (n == 0) return MMF_OK;
  {
    const bool overlap = exclude_self && (row_offset < col_offset + m) && (row_offset + n > col_offset);
    if (k > m - (overlap ? 1 : 0)) { set_error("simtopk_prepared: k exceeds the admissible columns"); return MMF_E_INVALID; }
  }
  const int kk = k + (exclude_self ? 1 : 0);
  if (!scan_bf16_supported(d, kk, in_dtype)) { set_error("simtopk_prepared: d = %lld / k = %d not supported by the 16-bit scan", (long long)d, k); return MMF_E_UNSUPPORTED; }
  const int cap = scan_f32_cap(kk);
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }
  FastTail ft(n, m, kk, cap, opts ? opts->col_splits : 0, scan_bf16_dp(d));
  MMF_TRY(ft.set_query_order(opts ? opts->query_order : MMF_QUERY_ORDER_AUTO));
  Workspace ws;
  MMF_TRY(get_workspace(device_id, s, ft.bytes(), &ws));
  ft.carve(ws);
  FastOperands fo{static_cast<const uint16_t*>(q->Z), static_cast<const uint16_t*>(c->Z), q->scal, c->scal, q->zn, q->rn, q->un,
                  c->cb, reinterpret_cast<const uint32_t*>(maxima), m_pad, scan_bf16_dp(d), operand == MMF_F16};
  return ft.run(X, n, Y, m, d, in_dtype, metric, lambda, k, exclude_self, row_offset, col_offset, fo, out_idx, out_val,
                opts && opts->profile, opts ? opts->select_wait_event : nullptr, stats,
                operand == MMF_F16 ? MMF_PREC_FAST : MMF_PREC_FAST_BF16, s);
}

int mmf_simtopk_panels(const void* X, int64_t n, const void* Y, int64_t m, int64_t d, int in_dtype, int metric,
                       float lambda, int k, int exclude_self, int64_t row_offset, int64_t col_offset,
                       const mmf_prepared_side* q, const float* c_scal, const mmf_panel* panels, int n_panels,
                       const float* maxima, int operand, int64_t* out_idx, float* out_val, const mmf_simtopk_opts* opts,
                       mmf_simtopk_stats* stats, int device_id, void* hip_stream) {
  MMF_TRY(check_common(X, n, m, d, in_dtype, device_id));
  if (!Y || !q || !c_scal || !panels || !maxima) { set_error("simtopk_panels: NULL pointer"); return MMF_E_INVALID; }
  if (metric < MMF_DOT || metric > MMF_RBF) { set_error("simtopk_panels: bad metric %d", metric); return MMF_E_INVALID; }
  if (metric == MMF_RBF && !(lambda > 0.0f)) { set_error("simtopk_panels: MMF_RBF needs lambda > 0"); return MMF_E_INVALID; }
  if (operand != MMF_F16 && operand != MMF_BF16) { set_error("simtopk_panels: bad operand"); return MMF_E_INVALID; }
  if (k < 1) { set_error("simtopk_panels: k must be >= 1"); return MMF_E_INVALID; }
  if (n_panels < 1 || n_panels > 16) { set_error("simtopk_panels: n_panels must be in 1..16"); return MMF_E_INVALID; }
  int64_t covered = 0, m_min = m, m_max = 0;
  for (int p = 0; p < n_panels; ++p) {
    const mmf_panel& P = panels[p];
    if (!P.Z || !P.cb || P.m < 1 || P.m_pad < P.m || (P.m_pad % 256) != 0 || P.seg_len < 0 || P.id_base < 0 ||
        (P.seg_len > 0 && (P.seg_stride < P.seg_len || (P.m % P.seg_len) != 0))) {
      set_error("simtopk_panels: panel %d is malformed", p); return MMF_E_INVALID;
    }
    const int64_t last = P.seg_len ? P.id_base + (P.m / P.seg_len - 1) * P.seg_stride + P.seg_len - 1 : P.id_base + P.m - 1;
    if (last >= m) { set_error("simtopk_panels: panel %d maps past column %lld", p, (long long)m); return MMF_E_INVALID; }
    covered += P.m;
    if (P.m < m_min) m_min = P.m;
    if (P.m_pad > m_max) m_max = P.m_pad;
  }
  if (covered != m) { set_error("simtopk_panels: panels cover %lld columns, Y has %lld", (long long)covered, (long long)m); return MMF_E_INVALID; }
  if (stats) memset(stats, 0, sizeof(*stats));
  if (n == 0) return MMF_OK;
  {
    const bool overlap = exclude_self && (row_offset < col_offset + m) && (row_offset + n > col_offset);
    if (k > m - (overlap ? 1 : 0)) { set_error("simtopk_panels: k exceeds the admissible columns"); return MMF_E_INVALID; }
  }
  const int kk = k + (exclude_self ? 1 : 0);
  if (!scan_bf16_supported(d, kk, in_dtype)) { set_error("simtopk_panels: d = %lld / k = %d not supported by the 16-bit scan", (long long)d, k); return MMF_E_UNSUPPORTED; }
  const int cap = scan_f32_cap(kk);
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }
  FastTail ft(n, m, kk, cap, opts ? opts->col_splits : 0, scan_bf16_dp(d), n_panels, m_min, m_max);
  MMF_TRY(ft.set_query_order(opts ? opts->query_order : MMF_QUERY_ORDER_AUTO));
  if ((int64_t)ft.lists * ft.bcap + FastTail::kSpillCap > 1024) {
    set_error("simtopk_panels: %d panels x %d-entry lists exceed the 1024 candidates a row can hand to the re-rank (k = %d): use fewer panels", n_panels, ft.bcap, k);
    return MMF_E_UNSUPPORTED;
  }
  Workspace ws;
  MMF_TRY(get_workspace(device_id, s, ft.bytes(), &ws));
  ft.carve(ws);
  FastOperands fo{static_cast<const uint16_t*>(q->Z), nullptr, q->scal, c_scal, q->zn, q->rn, q->un,
                  nullptr, reinterpret_cast<const uint32_t*>(maxima), 0, scan_bf16_dp(d), operand == MMF_F16};
  fo.panels = panels; fo.n_panels = n_panels;
  return ft.run(X, n, Y, m, d, in_dtype, metric, lambda, k, exclude_self, row_offset, col_offset, fo, out_idx, out_val,
                opts && opts->profile, opts ? opts->select_wait_event : nullptr, stats,
                operand == MMF_F16 ? MMF_PREC_FAST : MMF_PREC_FAST_BF16, s);
}

int mmf_topk_merge(const int64_t* ia, const float* va, const int64_t* ib, const float* vb, int64_t n, int k,
                   int64_t* io, float* vo, int device_id, void* hip_stream) {
  if (device_id < 0) { set_error("topk_merge: no CPU path"); return MMF_E_UNSUPPORTED; }
  if (n < 0 || k < 1) { set_error("topk_merge: bad n/k"); return MMF_E_INVALID; }
  if (n == 0) return MMF_OK;
  if (!ia || !va || !ib || !vb || !io || !vo) { set_error("topk_merge: NULL pointer"); return MMF_E_INVALID; }
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }
  return launch_topk_merge(ia, va, ib, vb, n, k, io, vo, static_cast<hipStream_t>(hip_stream));
}

int mmf_edge_cosine(const void* X, int64_t n, int64_t d, int in_dtype, const int64_t* edge_index, int64_t E,
                    float* out_w, int device_id, void* hip_stream) {
  MMF_TRY(check_common(X, n, n, d, in_dtype, device_id));
  if (E < 0) { set_error("edge_cosine: E < 0"); return MMF_E_INVALID; }
  if (E == 0) return MMF_OK;
  if (!edge_index || !out_w) { set_error("edge_cosine: NULL pointer"); return MMF_E_INVALID; }
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }
  Workspace ws;
  MMF_TRY(get_workspace(device_id, s, ws_bytes(n, 4), &ws));
  float* nrm = ws.take<float>(n);
  MMF_TRY(launch_row_scalars(X, n, d, in_dtype, MMF_COSINE, nrm, nullptr, s));
  return launch_edge_cosine_impl(X, d, in_dtype, nrm, edge_index, E, out_w, s);
}

int mmf_sim_dense(const void* X, int64_t n, const void* Y, int64_t m, int64_t d, int in_dtype, int metric,
                  float lambda, float* out, int device_id, void* hip_stream) {
  if (!Y) { Y = X; m = n; }
  MMF_TRY(check_common(X, n, m, d, in_dtype, device_id));
  if (metric < MMF_DOT || metric > MMF_RBF_DIRECT) { set_error("sim_dense: bad metric %d", metric); return MMF_E_INVALID; }
  if (n == 0 || m == 0) return MMF_OK;
  if (!out) { set_error("sim_dense: NULL output"); return MMF_E_INVALID; }
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }
  const bool same = (Y == X) && (m == n);
  float *rx = nullptr, *cy = nullptr, *Xp = nullptr, *Yp = nullptr;
  if (metric != MMF_RBF_DIRECT) {
    const bool img = sim_dense_needs_images(d, metric);
    Workspace ws;
    MMF_TRY(get_workspace(device_id, s, ws_bytes(n, 4) + ws_bytes(m, 4) +
                          (img ? ws_bytes(prep_f32_bytes(m, d), 1) + (same ? 0 : ws_bytes(prep_f32_bytes(n, d), 1)) : 0), &ws));
    rx = ws.take<float>(n);
    cy = same ? rx : ws.take<float>(m);
    MMF_TRY(launch_row_scalars(X, n, d, in_dtype, metric, rx, nullptr, s));
    if (!same) MMF_TRY(launch_row_scalars(Y, m, d, in_dtype, metric, cy, nullptr, s));
    if (img) {
      Yp = reinterpret_cast<float*>(ws.take<char>(prep_f32_bytes(m, d)));
      Xp = same ? Yp : reinterpret_cast<float*>(ws.take<char>(prep_f32_bytes(n, d)));
      MMF_TRY(launch_prep_f32(Y, m, d, in_dtype, nullptr, Yp, s));
      if (!same) MMF_TRY(launch_prep_f32(X, n, d, in_dtype, nullptr, Xp, s));
    }
  }
  return launch_sim_dense(X, n, Y, m, d, in_dtype, metric, lambda, rx, cy, Xp, Yp, out, s);
}

int mmf_sim_dense_stats(const void* X, int64_t n, const void* Y, int64_t m, int64_t d, int in_dtype, int metric, float lambda,
                        float* out, double* out_stats, int64_t panel_rows, int device_id, void* hip_stream) {
  if (!Y) { Y = X; m = n; }
  MMF_TRY(check_common(X, n, m, d, in_dtype, device_id));
  if (metric < MMF_DOT || metric > MMF_RBF_DIRECT) { set_error("sim_dense_stats: bad metric %d", metric); return MMF_E_INVALID; }
  if (n < 1 || m < 1) { set_error("sim_dense_stats: empty matrix"); return MMF_E_INVALID; }
  if (!out_stats) { set_error("sim_dense_stats: NULL out_stats"); return MMF_E_INVALID; }
  if (!out && metric != MMF_RBF_DIRECT) { set_error("sim_dense_stats: out == NULL is supported for MMF_RBF_DIRECT only"); return MMF_E_UNSUPPORTED; }
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const int64_t count = n * m;
  if (metric != MMF_RBF_DIRECT) {        // matrix-core dense kernels, then one reduction pass + the radix select
    MMF_TRY(mmf_sim_dense(X, n, Y, m, d, in_dtype, metric, lambda, out, device_id, hip_stream));
    return mmf_array_stats(out, count, out_stats, device_id, hip_stream);
  }
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }
  int64_t R = n;
  if (!out) {                            // nothing stored: rows recomputed in panels for each radix pass
    R = panel_rows > 0 ? panel_rows : (int64_t(1) << 30) / (4 * m);
    R = (R + 127) / 128 * 128;
    if (R < 128) R = 128;
    if (R > n) R = n;
  }
  int64_t blocks = 0;
  for (int64_t r0 = 0; r0 < n; r0 += R) blocks += rbf_direct_blocks((n - r0 < R) ? (n - r0) : R, m);
  const size_t mneed = median_scratch_bytes((unsigned long long)count);
  Workspace ws;
  MMF_TRY(get_workspace(device_id, s, ws_bytes((size_t)blocks * stat_partial_bytes(), 1) + ws_bytes(64, 4) + ws_bytes(mneed, 1) +
                                      (out ? 0 : ws_bytes((size_t)R * m, 4)), &ws));
  char* part = ws.take<char>((size_t)blocks * stat_partial_bytes());
  float* pivot = ws.take<float>(64);
  float* med = pivot + 8;
  void* mscratch = ws.take<char>(mneed);
  MMF_TRY(launch_rbf_direct_pivot(X, Y, d, in_dtype, lambda, pivot, s));
  if (out) {
    MMF_TRY(launch_rbf_direct(X, n, Y, m, d, in_dtype, lambda, out, part, pivot, s));
    MMF_TRY(launch_lower_median(out, count, med, mscratch, s));
  } else {
    // the matrix is never stored: every sweep of the median recomputes it panel by panel (one sweep when the sampled
    // bracket holds, four radix passes otherwise); the first sweep also leaves the statistic partials
    float* panel = ws.take<float>((size_t)R * m);
    bool partials_done = false;
    const MedianSweep sweep = [&](const MedianConsume& consume) -> int {
      int64_t b0 = 0;
      for (int64_t r0 = 0; r0 < n; r0 += R) {
        const int64_t rows = (n - r0 < R) ? (n - r0) : R;
        const void* Xr = static_cast<const char*>(X) + (size_t)r0 * d * dtype_size(in_dtype);
        MMF_TRY(launch_rbf_direct(Xr, rows, Y, m, d, in_dtype, lambda, panel,
                                  partials_done ? nullptr : part + (size_t)b0 * stat_partial_bytes(), pivot, s));
        MMF_TRY(consume(panel, m, median_no_diagonal_row(), rows));
        b0 += rbf_direct_blocks(rows, m);
      }
      partials_done = true;
      return MMF_OK;
    };
    MMF_TRY(lower_median_of((unsigned long long)count,
                            [&](float* sample, int sc) {
                              return launch_sample_pairs(X, Y, m, d, in_dtype, lambda, nullptr, 0, 0.0f, 0, (unsigned long long)count, sample, sc, s);
                            },
                            sweep, med, mscratch, s));
  }
  MMF_TRY(launch_stats_finish(part, blocks, pivot, count, out_stats, s));
  return launch_stats_set_median(med, out_stats, s);
}

int mmf_sim_dense_combined(const float* F, const float* P, int64_t n, int64_t d, int64_t dp, float lambda_h,
                           float lambda_g, float* out, int device_id, void* hip_stream) {
  MMF_TRY(check_common(F, n, n, d, MMF_F32, device_id));
  if (dp < 1) { set_error("sim_dense_combined: dp < 1"); return MMF_E_INVALID; }
  if (n == 0) return MMF_OK;
  if (!P || !out) { set_error("sim_dense_combined: NULL pointer"); return MMF_E_INVALID; }
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }
  Workspace ws;
  MMF_TRY(get_workspace(device_id, s, ws_bytes(n, 4) + ws_bytes(prep_f32_bytes(n, d), 1), &ws));
  float* nf = ws.take<float>(n);
  float* Fp = reinterpret_cast<float*>(ws.take<char>(prep_f32_bytes(n, d)));
  MMF_TRY(launch_row_scalars(F, n, d, MMF_F32, MMF_RBF, nf, nullptr, s));
  MMF_TRY(launch_prep_f32(F, n, d, MMF_F32, nullptr, Fp, s));
  return launch_sim_dense_combined(Fp, P, n, d, dp, lambda_h, lambda_g, nf, 0, n, out, s);
}

int mmf_offdiag_lower_median(const float* K, int64_t n, float* out_median, int device_id, void* hip_stream) {
  if (device_id < 0) { set_error("offdiag_lower_median: no CPU path"); return MMF_E_UNSUPPORTED; }
  if (n < 2) { set_error("offdiag_lower_median: need n >= 2 (got %lld)", (long long)n); return MMF_E_INVALID; }
  if (!K || !out_median) { set_error("offdiag_lower_median: NULL pointer"); return MMF_E_INVALID; }
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }
  const size_t need = median_scratch_bytes((unsigned long long)n * (unsigned long long)(n - 1));
  Workspace ws;
  MMF_TRY(get_workspace(device_id, s, ws_bytes(need, 1), &ws));
  return launch_offdiag_lower_median(K, n, out_median, ws.take<char>(need), s);
}

// ---- cluster-shaped steps (mmf_segments.hip) ----------------------------------------------------------------------
static int seg_common(const char* what, int64_t n, int64_t S, int device_id) {
  if (device_id < 0) { set_error("%s: no CPU path", what); return MMF_E_UNSUPPORTED; }
  if (n < 0 || S < 1) { set_error("%s: bad n / n_segments", what); return MMF_E_INVALID; }
  if (S > segment_max_segments()) { set_error("%s: at most %d segments are supported (got %lld)", what, segment_max_segments(), (long long)S); return MMF_E_UNSUPPORTED; }
  return MMF_OK;
}

int mmf_segment_sort(const int64_t* labels, int64_t n, int64_t n_segments, int64_t* counts, int64_t* offsets, int64_t* order,
                     int device_id, void* hip_stream) {
  MMF_TRY(seg_common("segment_sort", n, n_segments, device_id));
  if (!counts || !offsets || (n > 0 && (!labels || !order))) { set_error("segment_sort: NULL pointer"); return MMF_E_INVALID; }
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }
  if (n == 0) {
    MMF_HIP(hipMemsetAsync(counts, 0, (size_t)n_segments * 8, s));
    MMF_HIP(hipMemsetAsync(offsets, 0, (size_t)(n_segments + 1) * 8, s));
    return MMF_OK;
  }
  Workspace ws;
  MMF_TRY(get_workspace(device_id, s, ws_bytes(segment_sort_scratch_bytes(n, n_segments), 1) + ws_bytes(4, 4), &ws));
  char* scratch = ws.take<char>(segment_sort_scratch_bytes(n, n_segments));
  uint32_t* bad = ws.take<uint32_t>(4);
  MMF_TRY(launch_segment_sort(labels, n, n_segments, counts, offsets, order, scratch, bad, s));
  uint32_t h_bad = 0;
  MMF_HIP(hipMemcpyAsync(&h_bad, bad, 4, hipMemcpyDeviceToHost, s));
  MMF_HIP(hipStreamSynchronize(s));
  if (h_bad) { set_error("segment_sort: %u labels lie outside [0, %lld)", h_bad, (long long)n_segments); return MMF_E_INVALID; }
  return MMF_OK;
}

int mmf_segment_mean(const float* X, int64_t n, int64_t d, const int64_t* order, const int64_t* offsets, int64_t n_segments,
                     float* out, int device_id, void* hip_stream) {
  MMF_TRY(seg_common("segment_mean", n, n_segments, device_id));
  if (d < 1) { set_error("segment_mean: d < 1"); return MMF_E_INVALID; }
  if (!X || !order || !offsets || !out) { set_error("segment_mean: NULL pointer"); return MMF_E_INVALID; }
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }
  return launch_segment_mean(X, d, order, offsets, n_segments, out, static_cast<hipStream_t>(hip_stream));
}

int mmf_segment_offdiag_mean(const float* K, int64_t n, const int64_t* order, const int64_t* offsets, int64_t n_segments,
                             double* out_mean, int device_id, void* hip_stream) {
  MMF_TRY(seg_common("segment_offdiag_mean", n, n_segments, device_id));
  if (n < 1 || !K || !order || !offsets || !out_mean) { set_error("segment_offdiag_mean: NULL pointer / empty matrix"); return MMF_E_INVALID; }
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }
  Workspace ws;
  MMF_TRY(get_workspace(device_id, s, segment_offdiag_scratch_bytes(n), &ws));
  return launch_segment_offdiag_mean(K, n, order, offsets, n_segments, out_mean, ws.take<char>(segment_offdiag_scratch_bytes(n)), s);
}

int mmf_clique_pairs(const int64_t* order, const int64_t* offsets, int64_t n, int64_t n_segments, int64_t* pair_lo,
                     int64_t* pair_hi, int64_t capacity, int64_t* out_count, int device_id, void* hip_stream) {
  MMF_TRY(seg_common("clique_pairs", n, n_segments, device_id));
  if (capacity < 0 || !offsets || !out_count || (n > 0 && !order) || (capacity > 0 && (!pair_lo || !pair_hi))) {
    set_error("clique_pairs: NULL pointer / bad capacity"); return MMF_E_INVALID;
  }
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }
  Workspace ws;
  MMF_TRY(get_workspace(device_id, s, clique_scratch_bytes(n, n_segments), &ws));
  return launch_clique_pairs(order, offsets, n, n_segments, pair_lo, pair_hi, capacity, out_count,
                             ws.take<char>(clique_scratch_bytes(n, n_segments)), s);
}

int mmf_knn_pairs(const int64_t* nbr, int64_t n, int k, const int64_t* labels, int64_t* pair_lo, int64_t* pair_hi,
                  int64_t* out_count, int device_id, void* hip_stream) {
  if (device_id < 0) { set_error("knn_pairs: no CPU path"); return MMF_E_UNSUPPORTED; }
  if (n < 0 || k < 1) { set_error("knn_pairs: bad n / k"); return MMF_E_INVALID; }
  if (!out_count || (n > 0 && (!nbr || !pair_lo || !pair_hi))) { set_error("knn_pairs: NULL pointer"); return MMF_E_INVALID; }
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }
  return launch_knn_pairs(nbr, n, k, labels, pair_lo, pair_hi, out_count, static_cast<hipStream_t>(hip_stream));
}

int mmf_kmeans_fit(const float* X, int64_t n, int64_t d, int64_t n_clusters, int64_t n_init, int trials, const int64_t* first_centres,
                   const double* uniforms, int max_iter, double tol, int64_t* labels, float* centres, int64_t* seeds, double* info,
                   int device_id, void* hip_stream) {
  if (device_id < 0) { set_error("kmeans_fit: no CPU path"); return MMF_E_UNSUPPORTED; }
  if (n < 1 || d < 1 || n_clusters < 1 || n_clusters > n || n_init < 1 || trials < 1 || trials > 64 || max_iter < 1 || !(tol >= 0.0)) {
    set_error("kmeans_fit: need 1 <= n_clusters <= n, n_init >= 1, 1 <= trials <= 64, max_iter >= 1, tol >= 0 (n = %lld, n_clusters = %lld, "
              "n_init = %lld, trials = %d, max_iter = %d)", (long long)n, (long long)n_clusters, (long long)n_init, trials, max_iter);
    return MMF_E_INVALID;
  }
  if (n_init * n_clusters > segment_max_segments()) {
    set_error("kmeans_fit: n_init * n_clusters = %lld above the supported %d", (long long)(n_init * n_clusters), segment_max_segments());
    return MMF_E_UNSUPPORTED;
  }
  if (n_init * n >= ((int64_t)1 << 31)) { set_error("kmeans_fit: n_init * n must be < 2^31"); return MMF_E_UNSUPPORTED; }
  if (!X || !first_centres || (n_clusters > 1 && !uniforms) || !labels) { set_error("kmeans_fit: NULL pointer"); return MMF_E_INVALID; }
  for (int64_t i = 0; i < n_init; ++i)
    if (first_centres[i] < 0 || first_centres[i] >= n) { set_error("kmeans_fit: first_centres[%lld] outside [0, n)", (long long)i); return MMF_E_INVALID; }
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }
  const size_t need = kmeans_scratch_bytes(n, d, n_clusters, n_init, trials);
  Workspace ws;
  MMF_TRY(get_workspace(device_id, s, ws_bytes(need, 1), &ws));
  return launch_kmeans_fit(X, n, d, n_clusters, n_init, trials, first_centres, uniforms, max_iter, tol, labels, centres, seeds, info,
                           ws.take<char>(need), s);
}

int mmf_lower_median(const float* v, int64_t count, float* out_median, int device_id, void* hip_stream) {
  if (device_id < 0) { set_error("lower_median: no CPU path"); return MMF_E_UNSUPPORTED; }
  if (count < 1) { set_error("lower_median: need count >= 1 (got %lld)", (long long)count); return MMF_E_INVALID; }
  if (!v || !out_median) { set_error("lower_median: NULL pointer"); return MMF_E_INVALID; }
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }
  const size_t need = median_scratch_bytes((unsigned long long)count);
  Workspace ws;
  MMF_TRY(get_workspace(device_id, s, ws_bytes(need, 1), &ws));
  return launch_lower_median(v, count, out_median, ws.take<char>(need), s);
}

int mmf_array_stats(const float* v, int64_t count, double* out_stats, int device_id, void* hip_stream) {
  if (device_id < 0) { set_error("array_stats: no CPU path"); return MMF_E_UNSUPPORTED; }
  if (count < 1) { set_error("array_stats: need count >= 1 (got %lld)", (long long)count); return MMF_E_INVALID; }
  if (!v || !out_stats) { set_error("array_stats: NULL pointer"); return MMF_E_INVALID; }
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }
  Workspace ws;
  MMF_TRY(get_workspace(device_id, s, ws_bytes(array_stats_scratch_bytes(count), 1), &ws));
  return launch_array_stats(v, count, out_stats, ws.take<char>(array_stats_scratch_bytes(count)), s);
}

int mmf_threshold_edges(const float* K, int64_t n, float threshold, int64_t* edge_index, float* edge_w,
                        int64_t capacity, int64_t* out_count, int device_id, void* hip_stream) {
  if (device_id < 0) { set_error("threshold_edges: no CPU path"); return MMF_E_UNSUPPORTED; }
  if (n < 0 || capacity < 0) { set_error("threshold_edges: bad n/capacity"); return MMF_E_INVALID; }
  if (!out_count) { set_error("threshold_edges: NULL out_count"); return MMF_E_INVALID; }
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }
  MMF_HIP(hipMemsetAsync(out_count, 0, 8, s));
  if (n == 0) return MMF_OK;
  if (!K || (capacity > 0 && (!edge_index || !edge_w))) { set_error("threshold_edges: NULL pointer"); return MMF_E_INVALID; }
  const size_t rows_u32 = (size_t)n * 2 + 64;
  Workspace ws;
  MMF_TRY(get_workspace(device_id, s, ws_bytes(rows_u32, 8), &ws));
  return launch_threshold_edges(K, n, threshold, edge_index, edge_w, capacity, out_count,
                                reinterpret_cast<uint32_t*>(ws.take<uint64_t>(rows_u32)), rows_u32 * 2, s);
}

int mmf_threshold_edges_count(const float* K, int64_t n, float threshold, uint64_t* row_offsets, int64_t* out_count, int device_id,
                              void* hip_stream) {
  if (device_id < 0) { set_error("threshold_edges_count: no CPU path"); return MMF_E_UNSUPPORTED; }
  if (n < 0) { set_error("threshold_edges_count: bad n"); return MMF_E_INVALID; }
  if (!out_count || !row_offsets) { set_error("threshold_edges_count: NULL pointer"); return MMF_E_INVALID; }
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }
  MMF_HIP(hipMemsetAsync(out_count, 0, 8, s));
  if (n == 0) { MMF_HIP(hipMemsetAsync(row_offsets, 0, 8, s)); return MMF_OK; }
  if (!K) { set_error("threshold_edges_count: NULL pointer"); return MMF_E_INVALID; }
  Workspace ws;
  MMF_TRY(get_workspace(device_id, s, ws_bytes((size_t)n, 4), &ws));
  return launch_threshold_count(K, n, threshold, reinterpret_cast<unsigned long long*>(row_offsets), out_count, ws.take<uint32_t>((size_t)n), s);
}

int mmf_threshold_edges_fill(const float* K, int64_t n, float threshold, const uint64_t* row_offsets, int64_t* edge_index, float* edge_w,
                             int64_t capacity, int device_id, void* hip_stream) {
  if (device_id < 0) { set_error("threshold_edges_fill: no CPU path"); return MMF_E_UNSUPPORTED; }
  if (n < 0 || capacity < 0) { set_error("threshold_edges_fill: bad n/capacity"); return MMF_E_INVALID; }
  if (n == 0 || capacity == 0) return MMF_OK;
  if (!K || !row_offsets || !edge_index || !edge_w) { set_error("threshold_edges_fill: NULL pointer"); return MMF_E_INVALID; }
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }
  return launch_threshold_fill(K, n, threshold, reinterpret_cast<const unsigned long long*>(row_offsets), edge_index, edge_w, capacity,
                               static_cast<hipStream_t>(hip_stream));
}

// ---- the same two steps for an N whose K = K_h * K_g does not fit: K is recomputed in row panels -----------------
static int64_t pick_panel_rows(int64_t n, int64_t panel_rows) {
  if (panel_rows <= 0) panel_rows = (int64_t(1) << 30) / (4 * n);     // about 1 GiB of f32 per panel
  if (panel_rows < 128) panel_rows = 128;
  if (panel_rows > n) panel_rows = n;
  return panel_rows;
}

int mmf_combined_offdiag_median(const float* F, const float* P, int64_t n, int64_t d, int64_t dp, float lambda_h,
                                float lambda_g, int64_t panel_rows, float* out_median, int device_id, void* hip_stream) {
  MMF_TRY(check_common(F, n, n, d, MMF_F32, device_id));
  if (dp < 1 || n < 2) { set_error("combined_offdiag_median: need dp >= 1 and n >= 2"); return MMF_E_INVALID; }
  if (!P || !out_median) { set_error("combined_offdiag_median: NULL pointer"); return MMF_E_INVALID; }
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }
  const int64_t R = pick_panel_rows(n, panel_rows);
  Workspace ws;
  const unsigned long long count = (unsigned long long)n * (unsigned long long)(n - 1);
  const size_t mneed = median_scratch_bytes(count);
  MMF_TRY(get_workspace(device_id, s, ws_bytes(n, 4) + ws_bytes((size_t)R * n, 4) + ws_bytes(mneed, 1) +
                        ws_bytes(prep_f32_bytes(n, d), 1), &ws));
  float* nf = ws.take<float>(n);
  float* Kp = ws.take<float>((size_t)R * n);
  void* mscratch = ws.take<char>(mneed);
  float* Fp = reinterpret_cast<float*>(ws.take<char>(prep_f32_bytes(n, d)));
  MMF_TRY(launch_row_scalars(F, n, d, MMF_F32, MMF_RBF, nf, nullptr, s));
  MMF_TRY(launch_prep_f32(F, n, d, MMF_F32, nullptr, Fp, s));
  // one sweep over the recomputed matrix when the sampled bracket holds (four otherwise)
  return lower_median_of(
      count,
      [&](float* sample, int sc) { return launch_sample_pairs(F, F, n, d, MMF_F32, lambda_h, P, (int)dp, lambda_g, 1, count, sample, sc, s); },
      [&](const MedianConsume& consume) -> int {
        for (int64_t r0 = 0; r0 < n; r0 += R) {
          const int64_t rows = (n - r0 < R) ? (n - r0) : R;
          MMF_TRY(launch_sim_dense_combined(Fp, P, n, d, dp, lambda_h, lambda_g, nf, r0, rows, Kp, s));
          MMF_TRY(consume(Kp, n, r0, rows));
        }
        return MMF_OK;
      },
      out_median, mscratch, s);
}

int mmf_combined_threshold_edges(const float* F, const float* P, int64_t n, int64_t d, int64_t dp, float lambda_h,
                                 float lambda_g, float threshold, int64_t panel_rows, int64_t* edge_index, float* edge_w,
                                 int64_t capacity, int64_t* out_count, int device_id, void* hip_stream) {
  MMF_TRY(check_common(F, n, n, d, MMF_F32, device_id));
  if (dp < 1 || capacity < 0) { set_error("combined_threshold_edges: bad dp / capacity"); return MMF_E_INVALID; }
  if (!out_count) { set_error("combined_threshold_edges: NULL out_count"); return MMF_E_INVALID; }
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  DeviceGuard guard(device_id);
  if (!guard.ok) { set_error("hipSetDevice(%d) failed", device_id); return MMF_E_HIP; }
  MMF_HIP(hipMemsetAsync(out_count, 0, 8, s));
  if (n == 0) return MMF_OK;
  if (!P || (capacity > 0 && (!edge_index || !edge_w))) { set_error("combined_threshold_edges: NULL pointer"); return MMF_E_INVALID; }
  const int64_t R = pick_panel_rows(n, panel_rows);
  const size_t rows_u32 = (size_t)R * 2 + 64;
  Workspace ws;
  MMF_TRY(get_workspace(device_id, s, ws_bytes(n, 4) + ws_bytes((size_t)R * n, 4) + ws_bytes(rows_u32, 8) +
                        ws_bytes(prep_f32_bytes(n, d), 1), &ws));
  float* nf = ws.take<float>(n);
  float* Kp = ws.take<float>((size_t)R * n);
  uint32_t* scratch = reinterpret_cast<uint32_t*>(ws.take<uint64_t>(rows_u32));
  float* Fp = reinterpret_cast<float*>(ws.take<char>(prep_f32_bytes(n, d)));
  MMF_TRY(launch_row_scalars(F, n, d, MMF_F32, MMF_RBF, nf, nullptr, s));
  MMF_TRY(launch_prep_f32(F, n, d, MMF_F32, nullptr, Fp, s));
  for (int64_t r0 = 0; r0 < n; r0 += R) {            // panels in row order: the running count keeps the edges row-major
    const int64_t rows = (n - r0 < R) ? (n - r0) : R;
    MMF_TRY(launch_sim_dense_combined(Fp, P, n, d, dp, lambda_h, lambda_g, nf, r0, rows, Kp, s));
    MMF_TRY(launch_threshold_edges_panel(Kp, n, r0, rows, threshold, edge_index, edge_index ? edge_index + capacity : nullptr,
                                         edge_w, capacity, out_count, scratch, rows_u32 * 2, s));
  }
  return MMF_OK;
}

}  // extern "C"
