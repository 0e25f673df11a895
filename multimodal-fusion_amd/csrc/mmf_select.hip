// mmf_select.hip — exact re-rank of the scan's candidates and the final per-row top-k; plus the
// two small list kernels of the ABI (topk merge, per-edge cosine weights).
//
// select: one wave per query row.  Every candidate's key is recomputed as the canonical k-ordered
// fmaf chain from the ORIGINAL inputs (one lane per candidate walks k in order), self is dropped by
// identity, and k rounds of a wave-wide arg-best under (key desc, id asc) emit the row.  The result
// therefore does not depend on which scan kernel produced the candidates, on tile shapes, column
// splits or shard counts.
// Replaces: "indices[i, 1:]" (drop column 0) of preprocess_hypergraph.py:386-388 and the score
//           passes of similarity_kernel.py:49-52 for the pairs that survive.
#include <hipcub/hipcub.hpp>

#include "mmf_dev.h"
#include "mmf_host.h"

namespace mmf {

constexpr int SEL_WAVES = 4;
constexpr int SEL_MAXC = 1024;  // candidates per row the kernel can hold (lists * cap must fit)

struct SelectArgs {
  const void* X; const void* Y; int64_t n, m, d; int dtype;
  float neg_lambda; int k; int exclude_self; int64_t row_offset, col_offset;
  const float* rx; const float* cy;
  const int32_t* row_ids; int64_t n_rows;            // list position -> row of X (a subset of the rows, or a permutation: is_perm)
  int is_perm;
  const uint32_t* cand_cnt; const uint32_t* cand_ids; const uint32_t* overflow; int lists; int cap;
  const float* cand_keys; const float* margin;      // optional: approximate keys of the entries + the row's error margin
  int slot_ulp;                                     // ... which carry this many ulps of id-slot bits
  const uint32_t* spill_cnt; const uint32_t* spill_ids; int spill_cap;   // optional: the row's overflow list
  int spill_stacks;                                 // its two-stack form: spill_cnt[row] = front | back << 16 (SpillSink)
  // staged kernel, two passes when overflow lists exist: pass 0 handles the rows without overflow entries in a lean
  // LDS footprint and queues the others; pass 1 (room for the overflow entries) takes the queue
  int pass; int two_pass;
  // ordered second pass (near-duplicate data): pass 0 leaves a sort key per row — the smallest candidate id of a row that
  // waits for pass 1, 0xffffffff otherwise — and pass 1 walks the rows in key order, a contiguous eighth of that order per
  // XCD: rows whose candidate sets coincide (the members of a cluster) are re-ranked next to each other and find each
  // other's candidate rows in that XCD's L2 instead of gathering them from HBM again.
  uint32_t* key_in; uint32_t* row_in;              // pass 0 writes
  const uint32_t* key_sorted; const uint32_t* row_sorted;   // pass 1 reads
  uint8_t* done;                                   // [n_rows] rows the grouped re-rank (rerank_group_kernel) has finished
  uint8_t* late;                                   // [n_rows] rows the first pass handed on to the second (heavy after pruning), or nullptr
  int group_by_pos;                                // grouped re-rank: groups are 32 consecutive list positions (the scan took its queries with
                                                   //   near-duplicate rows next to each other: is_perm) instead of 32 rows of the key order
  int only_if;                                     // second pass: 0 always; 1 / 2: only when the grouped kernel finished less / not less than half of the waiting rows
  uint32_t* defer_cnt;                             // [512] rows that wait for the second pass (select_keys_kernel) and, behind them, rows the
                                                   //       grouped kernel finished — each count spread over 256 words
  int64_t order_blocks;                            // staged kernel: virtual blocks of SEL_WAVES rows the launch walks (grid-stride)
  int64_t* out_idx; float* out_val;
  int out_stride, out_off;                         // row stride of the outputs and first column this call fills
  float* floor_key_out; uint32_t* floor_id_out;    // optional: (key, local id) of the last entry emitted per row
  int32_t* fail_rows; uint32_t* fail_count; uint32_t* cand_total;
  int maxc;   // staged kernel: candidate slots per wave in dynamic LDS
};

template <bool VEC4>
__device__ __forceinline__ float chain_rows(const void* X, int64_t xr, const void* Y, int64_t yr, int64_t d,
                                            int dtype) {
  float acc = 0.0f;
  if constexpr (VEC4) {
    const f32x4* xp = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(X) + xr * d);
    const f32x4* yp = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(Y) + yr * d);
    const int64_t d4 = d >> 2;
    int64_t q = 0;
    // the chain is serial in k, the loads are not: fetch 8 x 16 B of each row ahead of 32 fmafs
    for (; q + 8 <= d4; q += 8) {
      f32x4 xv[8], yv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { xv[u] = xp[q + u]; yv[u] = yp[q + u]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc = __builtin_fmaf(xv[u][0], yv[u][0], acc);
        acc = __builtin_fmaf(xv[u][1], yv[u][1], acc);
        acc = __builtin_fmaf(xv[u][2], yv[u][2], acc);
        acc = __builtin_fmaf(xv[u][3], yv[u][3], acc);
      }
    }
    for (; q < d4; ++q) {
      const f32x4 xv = xp[q], yv = yp[q];
      acc = __builtin_fmaf(xv[0], yv[0], acc);
      acc = __builtin_fmaf(xv[1], yv[1], acc);
      acc = __builtin_fmaf(xv[2], yv[2], acc);
      acc = __builtin_fmaf(xv[3], yv[3], acc);
    }
  } else {
    for (int64_t k = 0; k < d; ++k)
      acc = __builtin_fmaf(ld_elem(X, xr * d + k, dtype), ld_elem(Y, yr * d + k, dtype), acc);
  }
  return acc;
}

// Does the row wait for the second pass (room for many candidates; grouped re-rank)?  When it has overflow entries — or when the first
// pass found it heavy: with many lists per row (paneled scans: a pair per panel) a cluster of near-duplicates fits the LISTS (18 lists hold
// 270 entries), the row has no overflow entry although it carries a hundred candidates after pruning, and the first pass hands it on
// (`late`; before pruning every row of a paneled scan looks heavy, so the raw counts cannot decide).
__device__ __forceinline__ bool row_waits(const SelectArgs& a, int64_t pos) {
  if (a.overflow[pos] != 0) return false;
  return a.spill_cnt[pos] != 0 || (a.late != nullptr && a.late[pos] != 0);
}

__device__ __forceinline__ int gather_candidates(const SelectArgs& a, int64_t pos, int lane, uint32_t* id, float* key,
                                                 int maxc) {
  int total = 0;
  const bool prune = a.cand_keys != nullptr && a.margin != nullptr;
  if (a.lists > 2 && a.lists <= 64 && a.cap <= 16) {
    // many short lists (paneled scans, column splits): one lane per list reads the counts, a wave scan places the lists, and four lists
    // are copied at a time (16 lanes each) — list after list, every count and every copy was a memory round trip of its own (18 lists:
    // 27 us per row).  Same layout as below: list 0's entries, then list 1's, ...
    const uint32_t mycn = lane < a.lists ? a.cand_cnt[pos * a.lists + lane] : 0u;
    uint32_t incl = mycn;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = (uint32_t)__shfl_up((int)incl, o);
      if (lane >= o) incl += t;
    }
    const int raw = __shfl((int)incl, 63);
    if (raw > maxc) return -1;
    const uint32_t excl = incl - mycn;
    const int g = lane >> 4, e = lane & 15;
    // (loads unconditional — a clamped address where the lane has no entry — and four steps unrolled: the loads of a step do not wait
    //  for the LDS writes of the one before)
#pragma unroll 4
    for (int l0 = 0; l0 < a.lists; l0 += 4) {
      const int l = l0 + g;
      const int src = l < a.lists ? l : 0;
      const uint32_t cn = (uint32_t)__shfl((int)mycn, src);
      const uint32_t off = (uint32_t)__shfl((int)excl, src);
      const bool mine = l < a.lists && (uint32_t)e < cn;
      const int64_t at = mine ? (pos * a.lists + l) * a.cap + e : pos * a.lists * a.cap;
      const uint32_t vi = a.cand_ids[at];
      const float vk = prune ? a.cand_keys[at] : 0.0f;
      if (mine) {
        id[off + e] = vi;
        if (prune) key[off + e] = vk;
      }
    }
    total = raw;
  } else
  for (int l = 0; l < a.lists; ++l) {
    const uint32_t cn = a.cand_cnt[pos * a.lists + l];
    const int64_t base = (pos * a.lists + l) * a.cap;
    if (total + (int)cn > maxc) return -1;
    for (uint32_t e = lane; e < cn; e += 64) {
      id[total + e] = a.cand_ids[base + e];
      if (prune) key[total + e] = a.cand_keys[base + e];
    }
    total += (int)cn;
  }
  // the row's overflow list (columns the lane lists had no room for): appended after the pruning below, which
  // works on approximate keys these entries do not carry
  // The counter counts RESERVED slots (lanes reserve in chunks, SpillSink): it may pass the capacity — a reservation
  // that found no room recorded its key as lost, and the audit flags the row if that key mattered — and reserved slots a
  // lane did not use hold 0xffffffff.
  int n_sp = 0, n_front = 0;
  if (a.spill_cnt) {
    const uint32_t c = a.spill_cnt[pos];
    if (a.spill_stacks) { n_front = (int)(c & 0xffffu); n_sp = n_front + (int)(c >> 16); if (n_sp > a.spill_cap) n_sp = a.spill_cap; }
    else { n_sp = (int)(c < (uint32_t)a.spill_cap ? c : (uint32_t)a.spill_cap); n_front = n_sp; }
  }
  auto append_spill = [&](int at) -> int {
    if (at + n_sp > maxc) return -1;
    for (int e0 = 0; e0 < n_sp; e0 += 64) {
      const int e = e0 + lane;
      // entry e: slot e of the front stack, or slot cap - 1 - (e - front) of the back stack
      const uint32_t v = (e < n_sp) ? a.spill_ids[pos * a.spill_cap + (e < n_front ? e : a.spill_cap - 1 - (e - n_front))] : 0xffffffffu;
      const bool keep = v != 0xffffffffu;
      const unsigned long long mask = __ballot(keep);
      if (keep) id[at + __popcll(mask & ((1ull << lane) - 1ull))] = v;
      at += __popcll(mask);
    }
    return at;
  };
  const int kk = a.k + (a.exclude_self ? 1 : 0);
  if (!prune || total <= kk) return append_spill(total);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  // t = kk-th largest approximate key: kk rounds of "best entry ranked after the previous pick" under
  // (key desc, position asc), so equal keys are counted once each
  float pk = 0.0f;
  uint32_t pe = 0;
  for (int t = 0; t < kk; ++t) {
    float bk = kNegInf;
    uint32_t be = kNoIdx;
    for (int e = lane; e < total; e += 64) {
      const float ke = key[e];
      if (t > 0 && !better(pk, pe, ke, (uint32_t)e)) continue;
      if (be == kNoIdx || better(ke, (uint32_t)e, bk, be)) { bk = ke; be = (uint32_t)e; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ok = __shfl_xor(bk, o);
      const uint32_t oe = (uint32_t)__shfl_xor((int)be, o);
      if (oe != kNoIdx && (be == kNoIdx || better(ok, oe, bk, be))) { bk = ok; be = oe; }
    }
    pk = bk; pe = be;
  }
  const float thr = pk - a.margin[pos];
  const int32_t tb = __float_as_int(thr);
  const int32_t thr_enc = tb >= 0 ? tb : (tb ^ 0x7fffffff);
  int kept = 0;
  for (int e0 = 0; e0 < total; e0 += 64) {             // in-place forward compaction (writes never pass reads)
    const int e = e0 + lane;
    bool keep = false;
    uint32_t ie = 0;
    if (e < total) {
      const int32_t b = __float_as_int(key[e]);
      keep = ((b >= 0 ? b : (b ^ 0x7fffffff)) + a.slot_ulp) >= thr_enc;
      ie = id[e];
    }
    __builtin_amdgcn_wave_barrier();
    const unsigned long long mask = __ballot(keep);
    if (keep) id[kept + __popcll(mask & ((1ull << lane) - 1ull))] = ie;
    kept += __popcll(mask);
  }
  return append_spill(kept);
}

template <int METRIC, bool VEC4>
__global__ __launch_bounds__(64 * SEL_WAVES) void select_kernel(SelectArgs a) {
  __shared__ float skey[SEL_WAVES][SEL_MAXC];
  __shared__ uint32_t sid[SEL_WAVES][SEL_MAXC];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t pos = (int64_t)blockIdx.x * SEL_WAVES + wave;
  if (pos >= a.n_rows) return;
  const int64_t row = a.row_ids ? (int64_t)a.row_ids[pos] : pos;
  float* key = skey[wave];
  uint32_t* id = sid[wave];

  bool failed = a.overflow[pos] != 0;
  const bool was_overflow = failed;
  int total = 0;
  if (!failed) {
    total = gather_candidates(a, pos, lane, id, key, SEL_MAXC);
    if (total < 0) { failed = true; total = 0; }
  }
  const int64_t grow = a.row_offset + row;
  int valid = 0;
  if (!failed) {
    __builtin_amdgcn_wave_barrier();
    const float ri = a.rx[row];
    for (int e = lane; e < total; e += 64) {
      const uint32_t j = id[e];
      const bool self = a.exclude_self && (a.col_offset + (int64_t)j == grow);
      float kx = kNegInf;
      if (!self && (int64_t)j < a.m) {
        const float dot = chain_rows<VEC4>(a.X, row, a.Y, (int64_t)j, a.d, a.dtype);
        kx = key_from_dot<METRIC>(dot, ri, a.cy[j], a.neg_lambda);
        if (kx != kx) kx = kNegInf;  // NaN keys rank last (documented: undefined for non-finite inputs)
        ++valid;
      } else {
        id[e] = kNoIdx;
      }
      key[e] = kx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) valid += __shfl_xor(valid, o);
    if (valid < a.k) failed = true;
  }
  if (failed) {
    if (lane == 0) {
      const uint32_t slot = atomicAdd(a.fail_count, 1u);
      a.fail_rows[slot] = (int32_t)row;
      atomicAdd(a.fail_count + (was_overflow ? 1 : 2), 1u);   // reason counters (stats only)
    }
    return;
  }
  if (a.cand_total && lane == 0) atomicAdd(a.cand_total + (blockIdx.x & 255), (uint32_t)total);
  __builtin_amdgcn_wave_barrier();
  for (int t = 0; t < a.k; ++t) {
    float bk = kNegInf;
    uint32_t bi = kNoIdx;
    int be = -1;
    for (int e = lane; e < total; e += 64) {
      const uint32_t ie = id[e];
      if (ie != kNoIdx && (be < 0 || better(key[e], ie, bk, bi))) { bk = key[e]; bi = ie; be = e; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ok = __shfl_xor(bk, o);
      const uint32_t oi = (uint32_t)__shfl_xor((int)bi, o);
      const int oe = __shfl_xor(be, o);
      const bool take = (oe >= 0) && (be < 0 || better(ok, oi, bk, bi));
      if (take) { bk = ok; bi = oi; be = oe; }
    }
    // every lane now agrees on (bk, bi, be); the owner retires the entry
    if (be >= 0 && (be & 63) == lane) id[be] = kNoIdx;
    if (lane == 0) {
      a.out_idx[row * a.out_stride + a.out_off + t] = a.col_offset + (int64_t)bi;
      a.out_val[row * a.out_stride + a.out_off + t] = val_from_key<METRIC>(bk);
      if (a.floor_key_out && t == a.k - 1) { a.floor_key_out[pos] = bk; a.floor_id_out[pos] = bi; }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// f32, d % 4 == 0: the same selection with the candidate rows staged through LDS.  Per query the wave
// walks its candidates in groups of SG and k in chunks of 128: every global read is a coalesced
// 512-byte row segment (two candidates per wave instruction, SG / 2 instructions in flight), and the
// canonical k-ordered chains run out of LDS (row stride 132 floats: conflict-free ds_read_b128),
// one lane per candidate.  SG = 8 for the common launch (6.5 candidates per row on Gaussian data: half the LDS, 24
// instead of 16 waves per CU, re-rank 1.35 -> 1.04 ms at N = 262144), 16 for the launch that takes the rows with
// overflow-list entries (129 candidates per row on clustered data: fewer group rounds, 11.6 -> 10.7 ms), 32 since round 3
// (with the rows of that launch ordered by their smallest candidate id: 10.8 -> 8.9 ms; the launch is bound by LDS reads —
// two ds_read_b128 per four fmaf of a lane — not by the gathers, which is why neither change buys more).
constexpr int SC = 128;
constexpr int SLD = SC + 4;

// four consecutive elements of a row as f32 (exact upcasts): 16 bytes of f32, or 8 bytes of bf16 / f16
template <int DT>
__device__ __forceinline__ f32x4 ld4_row(const void* base, int64_t elem) {
  if constexpr (DT == MMF_F32) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + elem);
  } else {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + elem);
    f32x4 v;
    if constexpr (DT == MMF_BF16) {
      v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
      v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    } else {
      v[0] = (float)__builtin_bit_cast(_Float16, (uint16_t)(u.x & 0xffffu)); v[1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(u.x >> 16));
      v[2] = (float)__builtin_bit_cast(_Float16, (uint16_t)(u.y & 0xffffu)); v[3] = (float)__builtin_bit_cast(_Float16, (uint16_t)(u.y >> 16));
    }
    return v;
  }
}

// the same four elements as ld4_row, split into the memory operation and the conversion: a load whose result is converted at once is
// waited for at once, and a ring of loads in flight needs the loads alone
template <int DT> struct Raw4 { typedef uint2 type; };
template <> struct Raw4<MMF_F32> { typedef f32x4 type; };
template <int DT>
__device__ __forceinline__ typename Raw4<DT>::type ld4_raw(const void* base, int64_t elem) {
  if constexpr (DT == MMF_F32) return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + elem);
  else return *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + elem);
}
template <int DT>
__device__ __forceinline__ f32x4 cvt4_raw(const typename Raw4<DT>::type& u) {
  if constexpr (DT == MMF_F32) {
    return u;
  } else {
    f32x4 v;
    if constexpr (DT == MMF_BF16) {
      v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
      v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    } else {
      v[0] = (float)__builtin_bit_cast(_Float16, (uint16_t)(u.x & 0xffffu)); v[1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(u.x >> 16));
      v[2] = (float)__builtin_bit_cast(_Float16, (uint16_t)(u.y & 0xffffu)); v[3] = (float)__builtin_bit_cast(_Float16, (uint16_t)(u.y >> 16));
    }
    return v;
  }
}

// Sort keys of the ordered second pass, one wave per 64 rows: a row that waits for that pass (overflow-list entries, not flagged)
// gets its smallest candidate id, every other row the marker 0xffffffff; the waiting rows are counted (spread over 256 words).
// Kept out of the first pass's kernel, whose register count decides how many of its waves fit a SIMD.
__global__ __launch_bounds__(256) void select_keys_kernel(SelectArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  const int64_t mine = r0 + lane;
  bool deferred = false;
  if (mine < a.n_rows) deferred = row_waits(a, mine);
  unsigned long long todo = __ballot(deferred);
  uint32_t key = 0xffffffffu;
  const int nwait = __popcll(todo);
  while (todo) {
    const int src = __builtin_ctzll(todo);
    todo &= todo - 1;
    const int64_t pos = r0 + src;
    uint32_t mn = 0xffffffffu;
    for (int l = 0; l < a.lists; ++l) {
      const uint32_t cn = a.cand_cnt[pos * a.lists + l];
      for (uint32_t e = lane; e < cn; e += 64) { const uint32_t v = a.cand_ids[(pos * a.lists + l) * a.cap + e]; mn = v < mn ? v : mn; }
    }
    const uint32_t c = a.spill_cnt[pos];
    uint32_t nfr, nsp;
    if (a.spill_stacks) { nfr = c & 0xffffu; nsp = nfr + (c >> 16); if (nsp > (uint32_t)a.spill_cap) nsp = (uint32_t)a.spill_cap; }
    else { nsp = c < (uint32_t)a.spill_cap ? c : (uint32_t)a.spill_cap; nfr = nsp; }
    for (uint32_t e = lane; e < nsp; e += 64) {
      const uint32_t v = a.spill_ids[pos * a.spill_cap + (e < nfr ? e : (uint32_t)a.spill_cap - 1u - (e - nfr))];
      mn = v < mn ? v : mn;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t v = (uint32_t)__shfl_xor((int)mn, o); mn = v < mn ? v : mn; }
    if (mn == 0xffffffffu) mn = 0xfffffffeu;       // a waiting row always carries a key below the "not waiting" marker
    if (lane == src) key = mn;
  }
  if (mine < a.n_rows) { a.key_in[mine] = key; a.row_in[mine] = (uint32_t)mine; }
  if (lane == 0 && nwait) atomicAdd(a.defer_cnt + (blockIdx.x & 255), (uint32_t)nwait);
}

template <int METRIC, int DT, int SG>
__global__ __launch_bounds__(64 * SEL_WAVES) void select_staged_kernel(SelectArgs a) {
  // dynamic LDS: per wave [maxc] keys + [maxc] ids (maxc = lists * cap rounded up to 64), then the tiles
  extern __shared__ __attribute__((aligned(16))) char sel_smem[];
  const int maxc = a.maxc;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float (*ytile)[SG][SLD] = reinterpret_cast<float (*)[SG][SLD]>(sel_smem);
  float (*xtile)[SC] = reinterpret_cast<float (*)[SC]>(sel_smem + sizeof(float) * SEL_WAVES * SG * SLD);
  float* skey_base = reinterpret_cast<float*>(sel_smem + sizeof(float) * SEL_WAVES * (SG * SLD + SC));
  uint32_t* sid_base = reinterpret_cast<uint32_t*>(skey_base + SEL_WAVES * maxc);
  // The ordered second pass walks its (virtual) blocks with a grid-stride loop: behind the grouped re-rank almost every row is
  // already done, and 65536 workgroups that only look at a flag cost 1.5 ms in workgroup launches alone.
  constexpr bool SECOND = SG != 8;          // SG = 8 is the lean first pass: one virtual block per workgroup, no order
  if (SECOND && a.only_if) {                 // the launch shaped for the other case leaves at once
    uint32_t c = 0, dn = 0;
    for (int i = lane; i < 256; i += 64) { c += a.defer_cnt[i]; dn += a.defer_cnt[256 + i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { c += (uint32_t)__shfl_xor((int)c, o); dn += (uint32_t)__shfl_xor((int)dn, o); }
    const bool scattered = (int64_t)dn * 2 < (int64_t)c;     // the grouped kernel finished less than half of the waiting rows
    if ((a.only_if == 1) != scattered) return;
  }
  for (int64_t vb = blockIdx.x; vb < (SECOND ? a.order_blocks : (int64_t)gridDim.x); vb += gridDim.x) {
  int64_t pos = vb * SEL_WAVES + wave;
  if (SECOND && a.pass == 1 && a.key_sorted && a.only_if != 1) {   // (the launch for scattered rows walks them in row order: their
                                                                     //  candidate sets share nothing, and row order keeps the list reads streaming)
    // virtual block b -> chunk (b % 8) * (blocks / 8) + b / 8 of the sorted order (order_blocks is a multiple of 8)
    const int64_t per = a.order_blocks >> 3;
    const int64_t slot = ((vb & 7) * per + (vb >> 3)) * SEL_WAVES + wave;
    if (slot >= a.n_rows || a.key_sorted[slot] == 0xffffffffu) continue;
    pos = (int64_t)a.row_sorted[slot];
    if (a.done && a.done[pos]) continue;
  } else {
    if (pos >= a.n_rows) continue;
    if (a.two_pass) {
      // two passes: rows with overflow entries wait for the second launch, which has LDS room for them.  Both launches
      // cover every row and a row decides by its own counters which one it belongs to (a queue filled through one
      // atomic counter cost 25 cycles per row at the L2 when every row has overflow entries, i.e. on clustered data).
      const bool deferred = row_waits(a, pos);
      if (deferred != (a.pass == 1)) continue;
      if (SECOND && a.pass == 1 && a.done && a.done[pos]) continue;
    }
  }
  const int64_t row = a.row_ids ? (int64_t)a.row_ids[pos] : pos;
  float* key = skey_base + wave * maxc;
  uint32_t* id = sid_base + wave * maxc;
  const void* X = a.X;
  const void* Y = a.Y;

  bool failed = a.overflow[pos] != 0;
  const bool was_overflow = failed;
  int total = 0;
  if (!failed) {
    total = gather_candidates(a, pos, lane, id, key, maxc);
    if (total < 0) { failed = true; total = 0; }
  }
  if (!SECOND && a.two_pass && a.late && !failed && total > 2 * (a.k + a.exclude_self) + 32) {    // heavy after pruning: the second pass (grouped re-rank) takes it
    if (lane == 0) a.late[pos] = 1;
    continue;
  }
  const int64_t grow = a.row_offset + row;
  int valid = 0;
  if (!failed) {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    const float ri = a.rx[row];
    const int hl = lane >> 5, ll = lane & 31;
    for (int g0 = 0; g0 < total; g0 += SG) {
      const int ng = (total - g0 < SG) ? (total - g0) : SG;
      // this lane's candidate (chain phase) and its admissibility
      uint32_t myj = kNoIdx;
      bool ok = false;
      if (lane < ng) {
        myj = id[g0 + lane];
        ok = ((int64_t)myj < a.m) && !(a.exclude_self && (a.col_offset + (int64_t)myj == grow));
      }
      float acc = 0.0f;
      // loads of chunk c+1 are issued before the chains of chunk c run (registers), written to LDS after
      f32x4 xv, yv[SG / 2];
      auto gload = [&](int64_t k0) {
        const int kc = (a.d - k0 < SC) ? (int)(a.d - k0) : SC;
        xv = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (hl == 0 && 4 * ll < kc) xv = ld4_row<DT>(X, row * a.d + k0 + 4 * ll);
#pragma unroll
        for (int p = 0; p < SG / 2; ++p) {
          const int cnd = 2 * p + hl;
          yv[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (cnd < ng && 4 * ll < kc) {
            uint32_t j = id[g0 + cnd];
            if ((int64_t)j >= a.m) j = 0;
            yv[p] = ld4_row<DT>(Y, (int64_t)j * a.d + k0 + 4 * ll);
          }
        }
      };
      gload(0);
      for (int64_t k0 = 0; k0 < a.d; k0 += SC) {
        const int kc = (a.d - k0 < SC) ? (int)(a.d - k0) : SC;
        if (hl == 0) *reinterpret_cast<f32x4*>(&xtile[wave][4 * ll]) = xv;
#pragma unroll
        for (int p = 0; p < SG / 2; ++p) *reinterpret_cast<f32x4*>(&ytile[wave][2 * p + hl][4 * ll]) = yv[p];
        if (k0 + SC < a.d) gload(k0 + SC);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        if (lane < ng) {
          for (int kk = 0; kk < kc; kk += 4) {
            const f32x4 x4 = *reinterpret_cast<const f32x4*>(&xtile[wave][kk]);
            const f32x4 y4 = *reinterpret_cast<const f32x4*>(&ytile[wave][lane][kk]);
            acc = __builtin_fmaf(x4[0], y4[0], acc);
            acc = __builtin_fmaf(x4[1], y4[1], acc);
            acc = __builtin_fmaf(x4[2], y4[2], acc);
            acc = __builtin_fmaf(x4[3], y4[3], acc);
          }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
      }
      if (lane < ng) {
        float kx = kNegInf;
        if (ok) {
          kx = key_from_dot<METRIC>(acc, ri, a.cy[myj], a.neg_lambda);
          if (kx != kx) kx = kNegInf;
          ++valid;
        } else {
          id[g0 + lane] = kNoIdx;
        }
        key[g0 + lane] = kx;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) valid += __shfl_xor(valid, o);
    if (valid < a.k) failed = true;
  }
  if (failed) {
    if (lane == 0) {
      const uint32_t slot = atomicAdd(a.fail_count, 1u);
      a.fail_rows[slot] = (int32_t)row;
      atomicAdd(a.fail_count + (was_overflow ? 1 : 2), 1u);
    }
    continue;
  }
  if (a.cand_total && lane == 0) atomicAdd(a.cand_total + (blockIdx.x & 255), (uint32_t)total);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  if (total <= 64) {
    // One candidate per lane (the usual row: two lane lists): every lane COUNTS the candidates that rank before its own — each one
    // broadcast out of its lane's registers by v_readlane, no LDS trip, no butterfly — and the k lanes with ranks below k write the
    // row's output themselves.  (key desc, id asc) is a strict total order on distinct ids, so the ranks are distinct.
    const float mk = lane < total ? key[lane] : kNegInf;
    const uint32_t mi = lane < total ? id[lane] : kNoIdx;
    int rank = 0;
    for (int e = 0; e < total; ++e) {
      const float ke = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(mk), e));
      const uint32_t ie = (uint32_t)__builtin_amdgcn_readlane((int)mi, e);
      rank += (ie != kNoIdx && better(ke, ie, mk, mi)) ? 1 : 0;
    }
    if (mi != kNoIdx && rank < a.k) {
      a.out_idx[row * a.out_stride + a.out_off + rank] = a.col_offset + (int64_t)mi;
      a.out_val[row * a.out_stride + a.out_off + rank] = val_from_key<METRIC>(mk);
      if (a.floor_key_out && rank == a.k - 1) { a.floor_key_out[pos] = mk; a.floor_id_out[pos] = mi; }
    }
    continue;
  }
  for (int t = 0; t < a.k; ++t) {
    float bk = kNegInf;
    uint32_t bi = kNoIdx;
    int be = -1;
    for (int e = lane; e < total; e += 64) {
      const uint32_t ie = id[e];
      if (ie != kNoIdx && (be < 0 || better(key[e], ie, bk, bi))) { bk = key[e]; bi = ie; be = e; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ok2 = __shfl_xor(bk, o);
      const uint32_t oi = (uint32_t)__shfl_xor((int)bi, o);
      const int oe = __shfl_xor(be, o);
      const bool take = (oe >= 0) && (be < 0 || better(ok2, oi, bk, bi));
      if (take) { bk = ok2; bi = oi; be = oe; }
    }
    if (be >= 0 && (be & 63) == lane) id[be] = kNoIdx;
    if (lane == 0) {
      a.out_idx[row * a.out_stride + a.out_off + t] = a.col_offset + (int64_t)bi;
      a.out_val[row * a.out_stride + a.out_off + t] = val_from_key<METRIC>(bk);
      if (a.floor_key_out && t == a.k - 1) { a.floor_key_out[pos] = bk; a.floor_id_out[pos] = bi; }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
  }
  }   // virtual blocks
}

// ------------------------------------------------------------------------------------------------
// Grouped re-rank of the rows that wait for the second pass (near-duplicate data: a row's margin band is its whole cluster,
// 129 candidates per row at the benchmark's clustered workload).  In key order — smallest candidate id — 32 consecutive rows
// mostly share one candidate set, so a workgroup takes 32 such rows, forms the UNION of their candidates (an LDS hash table:
// id -> panel column), and computes all 32 x |union| canonical dots on the matrix cores: v_mfma_f32_32x32x2_f32 accumulates
// exactly the k-ordered fmaf chain (products commute), so the keys are the bits the per-row chains produce.
// Each row then looks its own candidates up and selects as before.  Candidate rows are gathered once per 32 queries instead
// of once per query, and no lane walks a 512-long chain out of LDS.  A group whose union exceeds GR_PMAX columns (or a row
// the lists could not serve) is left to the per-row pass behind this kernel (a.done stays 0).
// ------------------------------------------------------------------------------------------------
constexpr int GR_Q = 32, GR_PMAX = 512, GR_KC = 16, GR_MAXC = 224, GR_HASH = 1024, GR_LD = GR_KC + 1, GR_W = 8, GR_AHEAD = 8;
static_assert(GR_Q * (GR_PMAX + 1) >= GR_W * 2 * SEL_MAXC, "the dots' LDS doubles as the waves' gather scratch");
struct GroupLds {
  uint32_t cid[GR_Q][GR_MAXC];
  float dots[GR_Q][GR_PMAX + 1];       // before the matrix-core phase its first GR_Q x GR_MAXC floats hold the lists' approximate keys (gather_candidates' pruning)
  float qtile[2][GR_Q][GR_LD];
  float ptile[2][GR_W][32][GR_LD];
  uint32_t hkey[GR_HASH];
  uint32_t hval[GR_HASH];
  uint32_t panel[GR_PMAX];
  float pcy[GR_PMAX];
  float qrx[GR_Q];
  int qrow[GR_Q];
  int qpos[GR_Q];
  int qtot[GR_Q];
  int np;
  int fail;
  int first;
};

template <int METRIC, int DT>
__global__ __launch_bounds__(64 * GR_W) void rerank_group_kernel(SelectArgs a) {
  extern __shared__ __attribute__((aligned(16))) char grp_smem[];
  GroupLds& L = *reinterpret_cast<GroupLds*>(grp_smem);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // block b -> group (b % 8) * (groups / 8) + b / 8 of the sorted order: an XCD walks a contiguous eighth of it
  const int64_t per = gridDim.x >> 3;
  const int64_t grp = ((int64_t)blockIdx.x & 7) * per + ((int64_t)blockIdx.x >> 3);
  const int64_t slot0 = grp * GR_Q;
  if (slot0 >= a.n_rows) return;
  if (tid < GR_Q) {
    const int64_t slot = slot0 + tid;
    int p = -1;
    if (slot < a.n_rows) {
      if (a.group_by_pos) { if (row_waits(a, slot)) p = (int)slot; }
      else if (a.key_sorted[slot] != 0xffffffffu) p = (int)a.row_sorted[slot];
    }
    L.qpos[tid] = p;
    L.qtot[tid] = -1;
    const int r = p < 0 ? -1 : (a.row_ids ? (int)a.row_ids[p] : p);       // list position -> row of X, and the row's scalar: fetched once,
    L.qrow[tid] = r;                                                       // not as two dependent loads in front of every row's selection
    L.qrx[tid] = r < 0 ? 0.0f : a.rx[r];
  }
  for (int i = tid; i < GR_HASH; i += 64 * GR_W) L.hkey[i] = 0xffffffffu;
  if (tid == 0) { L.np = 0; L.fail = 0; }
  __syncthreads();
  if (tid < 64) {                                   // the rows of this group that wait, and the first of them (stand-in for the others)
    const unsigned long long wait = __ballot(tid < GR_Q && L.qpos[tid < GR_Q ? tid : 0] >= 0);
    if (tid == 0) {
      L.first = wait ? __builtin_ctzll(wait) : -1;
      // position order: nobody has counted the waiting rows yet (select_keys_kernel does it for the key order)
      if (a.group_by_pos && wait) atomicAdd(a.defer_cnt + (blockIdx.x & 255), (uint32_t)__popcll(wait));
    }
  }
  __syncthreads();
  if (L.first < 0) return;                          // nothing waits here (key order: nor from here on)
  // Grouping pays when the rows of a group share their candidates — near-duplicate rows, which then carry the SAME key (the
  // smallest id of their common candidate set), so a group holds a few runs of equal keys.  Scattered rows with overflow
  // entries (Gaussian rows under bf16 operands) carry 32 different keys and share nothing: their union would overflow the
  // panel, and finding that out costs more than the per-row pass.  A group with more than four runs of keys is left alone.
  {
    const bool change = !a.group_by_pos && tid > 0 && tid < GR_Q && L.qpos[tid] >= 0 && a.key_sorted[slot0 + tid] != a.key_sorted[slot0 + tid - 1];
    if (tid < 64) { const int cnt = __popcll(__ballot(change)); if (tid == 0) L.fail = cnt; }
  }
  __syncthreads();
  const bool alone = L.fail > 3;
  __syncthreads();
  if (alone) return;
  if (tid == 0) L.fail = 0;
  __syncthreads();
  // (1) the rows' candidate lists (pruned by the union of their lists as in the per-row pass)
  for (int qq = 0; qq < GR_Q / GR_W; ++qq) {
    const int q = (GR_Q / GR_W) * w + qq;
    const int pos = L.qpos[q];
    if (pos < 0) continue;
    int total = -1;
    if (a.overflow[pos] == 0) {
      if (a.lists * a.cap + a.spill_cap <= GR_MAXC) {
        total = gather_candidates(a, pos, lane, L.cid[q], &L.dots[0][0] + q * GR_MAXC, GR_MAXC);
      } else {
        // many lists per row (paneled scans: one list pair per panel): the RAW entries can exceed what a row keeps here although the
        // pruned set does not — gather and prune in this wave's 1024-entry scratch (the dots' LDS, idle until the matrix-core phase),
        // then keep the result if it fits
        uint32_t* sid = reinterpret_cast<uint32_t*>(&L.dots[0][0]) + w * (2 * SEL_MAXC);
        float* skey = reinterpret_cast<float*>(sid + SEL_MAXC);
        total = gather_candidates(a, pos, lane, sid, skey, SEL_MAXC);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        if (total > GR_MAXC) total = -1;
        for (int e = lane; e < total; e += 64) L.cid[q][e] = sid[e];
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (lane == 0) L.qtot[q] = total;
  }
  __syncthreads();
  // A group whose union does not fit the panel (rows of two or three clusters in one group) is taken as two halves of 16 rows, one
  // after the other — half-empty matrix-core tiles, but still far cheaper than 32 per-row passes; a half that does not fit either is
  // left to the per-row pass.
  int nparts = 1;
  bool fresh = true;
  for (int part = 0; part < nparts; ++part) {
  const int r0 = nparts == 1 ? 0 : (GR_Q / 2) * part, r1 = nparts == 1 ? GR_Q : r0 + GR_Q / 2;
  if (!fresh) {
    __syncthreads();
    for (int i = tid; i < GR_HASH; i += 64 * GR_W) L.hkey[i] = 0xffffffffu;
    if (tid == 0) { L.np = 0; L.fail = 0; }
    __syncthreads();
  }
  fresh = false;
  // (2) union of the candidates: id -> panel column.  Four rows at a time (128 threads each); the rows of a group mostly carry the
  // same ids, so an entry is usually found by a plain read and the compare-and-swap is left to the first row that brings an id.
  for (int it = 0; it < GR_Q / 4; ++it) {
    const int q = 4 * it + (tid >> 7);
    const int total = (q >= r0 && q < r1) ? L.qtot[q] : -1;
    for (int e = tid & 127; e < total; e += 128) {
      const uint32_t id = L.cid[q][e];
      if ((int64_t)id >= a.m) continue;
      uint32_t h = (id * 2654435761u) >> 22;
      for (int probe = 0; probe < GR_HASH; ++probe) {
        uint32_t prev = *(volatile uint32_t*)&L.hkey[h];
        if (prev == id) break;
        if (prev == 0xffffffffu) {
          if (*(volatile int*)&L.fail) break;
          prev = atomicCAS(&L.hkey[h], 0xffffffffu, id);
          if (prev == 0xffffffffu) {
            const int idx = atomicAdd(&L.np, 1);
            if (idx < GR_PMAX) { L.panel[idx] = id; L.hval[h] = (uint32_t)idx; } else L.fail = 1;
            break;
          }
          if (prev == id) break;
        }
        h = (h + 1) & (GR_HASH - 1);
      }
    }
  }
  __syncthreads();
  if (!L.fail) for (int i = tid; i < L.np; i += 64 * GR_W) L.pcy[i] = a.cy[L.panel[i]];     // the columns' scalars, one gather for the group
  __syncthreads();
  if (L.fail) {
    if (nparts == 1) { nparts = 2; part = -1; }     // again, as two halves
    continue;
  }
  const int np = L.np;
  // (3) all 32 x np canonical dots: in pass p wave w owns panel columns [256 p + 32 w, + 32)
  for (int pass = 0; pass * 32 * GR_W < np; ++pass) {
    const int64_t d = a.d;
    const int nchunk = (int)((d + GR_KC - 1) / GR_KC);
    const bool qstage = tid < 128;                                    // query tile: 32 rows x 16 k = 128 16-byte pieces
    const int sq_row = (tid >> 2) & 31, sq_k = (tid & 3) * 4;
    const int64_t qrow = L.qpos[sq_row] >= 0 ? (int64_t)L.qrow[sq_row] : (int64_t)L.qrow[L.first];
    const int c0 = pass * 32 * GR_W + 32 * w;
    const bool active = c0 < np;                                      // waves beyond the panel only keep the barriers
    int64_t prow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = c0 + (lane >> 2) + 16 * i;
      prow[i] = (int64_t)L.panel[c < np ? c : 0];
    }
    const int pk = (lane & 3) * 4;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    // Global loads run GR_AHEAD chunks ahead of the matrix cores, in a ring of register stages (stage = chunk % GR_AHEAD): a chunk is
    // 64 bytes of each gathered row, and with one chunk in flight every one of the d / 16 steps waited a full memory round trip.
    // The loads are UNCONDITIONAL (clamped addresses; what lies outside the rows' d elements or past the last chunk is zeroed when it
    // is written to LDS): a load under a branch, or one converted where it is issued, makes the compiler wait for everything in
    // flight — s_waitcnt vmcnt(0) — at the next LDS write, and the ring held one chunk instead of eight.
    typedef typename Raw4<DT>::type raw_t;
    raw_t qv[GR_AHEAD], pv[GR_AHEAD][2];
    auto gload = [&](int ch, raw_t& q_, raw_t (&p_)[2]) {
      const int64_t k0 = (int64_t)(ch < nchunk ? ch : nchunk - 1) * GR_KC;
      const int64_t kq = (k0 + sq_k < d) ? k0 + sq_k : d - 4, kp = (k0 + pk < d) ? k0 + pk : d - 4;
      q_ = ld4_raw<DT>(a.X, qrow * d + kq);
#pragma unroll
      for (int i = 0; i < 2; ++i) p_[i] = ld4_raw<DT>(a.Y, prow[i] * d + kp);
    };
    auto swrite = [&](int buf, int ch, const raw_t& q_, const raw_t (&p_)[2]) {
      const int64_t k0 = (int64_t)ch * GR_KC;
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      if (qstage) {
        const f32x4 qf = (ch < nchunk && k0 + sq_k < d) ? cvt4_raw<DT>(q_) : zero;
#pragma unroll
        for (int e = 0; e < 4; ++e) L.qtile[buf][sq_row][sq_k + e] = qf[e];
      }
      if (active) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const f32x4 pf = (ch < nchunk && k0 + pk < d) ? cvt4_raw<DT>(p_[i]) : zero;
#pragma unroll
          for (int e = 0; e < 4; ++e) L.ptile[buf][w][(lane >> 2) + 16 * i][pk + e] = pf[e];
        }
      }
    };
#pragma unroll
    for (int j = 0; j < GR_AHEAD; ++j) gload(j, qv[j], pv[j]);
    swrite(0, 0, qv[0], pv[0]);
    gload(GR_AHEAD, qv[0], pv[0]);
    __syncthreads();
    // (the loop runs whole rounds of GR_AHEAD steps: past the last chunk a step only writes zeros and issues a clamped load, so that
    //  NO memory operation sits under a branch and the compiler can wait for a stage with a counted vmcnt instead of vmcnt(0))
    for (int base = 0; base < nchunk; base += GR_AHEAD) {
#pragma unroll
      for (int j = 0; j < GR_AHEAD; ++j) {
        const int ch = base + j;                      // in LDS buffer ch & 1; chunk ch + 1 waits in stage (j + 1) % GR_AHEAD
        const int buf = ch & 1;
        if (active && ch < nchunk) {
          const float* qa = &L.qtile[buf][lane & 31][lane >> 5];
          const float* pb = &L.ptile[buf][w][lane & 31][lane >> 5];
#pragma unroll
          for (int sx = 0; sx < GR_KC / 2; ++sx) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[2 * sx], pb[2 * sx], acc, 0, 0, 0);
        }
        constexpr int GR_AH = GR_AHEAD;
        const int nx = (j + 1) % GR_AH;
        swrite(buf ^ 1, ch + 1, qv[nx], pv[nx]);
        gload(ch + 1 + GR_AHEAD, qv[nx], pv[nx]);
        __syncthreads();
      }
    }
    // C layout: column (candidate) = lane & 31, row (query) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    if (active) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (c0 + (lane & 31) < GR_PMAX) L.dots[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][c0 + (lane & 31)] = acc[r];
    }
  }
  __syncthreads();
  // (4) every row: keys of its own candidates out of the dots, then the top-k under (key desc, id asc)
  for (int qq = 0; qq < GR_Q / GR_W; ++qq) {
    const int q = (GR_Q / GR_W) * w + qq;
    const int pos = L.qpos[q];
    const int total = L.qtot[q];
    if (pos < 0 || total < 0 || q < r0 || q >= r1) continue;
    const uint32_t* id = L.cid[q];
    const int64_t row = (int64_t)L.qrow[q];
    const int64_t grow = a.row_offset + row;
    const float ri = L.qrx[q];
    // a lane keeps its (up to four) candidates in registers: exact key and id, kNoIdx once taken or inadmissible
    constexpr int GR_E = (GR_MAXC + 63) / 64;
    float kx[GR_E];
    uint32_t jx[GR_E];
    int valid = 0;
#pragma unroll
    for (int i = 0; i < GR_E; ++i) {
      const int e = lane + 64 * i;
      kx[i] = kNegInf;
      jx[i] = kNoIdx;
      if (e < total) {
        const uint32_t j = id[e];
        if ((int64_t)j < a.m && !(a.exclude_self && (a.col_offset + (int64_t)j == grow))) {
          uint32_t h = (j * 2654435761u) >> 22;
          while (L.hkey[h] != j) h = (h + 1) & (GR_HASH - 1);
          const uint32_t col = L.hval[h];
          float kv = key_from_dot<METRIC>(L.dots[q][col], ri, L.pcy[col], a.neg_lambda);
          if (kv != kv) kv = kNegInf;
          kx[i] = kv;
          jx[i] = j;
          ++valid;
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) valid += __shfl_xor(valid, o);
    if (valid < a.k) continue;                      // short of candidates: the per-row pass reports the row
    if (a.cand_total && lane == 0) atomicAdd(a.cand_total + (blockIdx.x & 255), (uint32_t)total);
    for (int t = 0; t < a.k; ++t) {
      float bk = kNegInf;
      uint32_t bi = kNoIdx;
      int bs = -1;                                  // lane + 64 * register slot of the best entry
#pragma unroll
      for (int i = 0; i < GR_E; ++i)
        if (jx[i] != kNoIdx && (bs < 0 || better(kx[i], jx[i], bk, bi))) { bk = kx[i]; bi = jx[i]; bs = lane + 64 * i; }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ok2 = __shfl_xor(bk, o);
        const uint32_t oi = (uint32_t)__shfl_xor((int)bi, o);
        const int os = __shfl_xor(bs, o);
        const bool take = (os >= 0) && (bs < 0 || better(ok2, oi, bk, bi));
        if (take) { bk = ok2; bi = oi; bs = os; }
      }
      if (bs >= 0 && (bs & 63) == lane) {
#pragma unroll
        for (int i = 0; i < GR_E; ++i)
          if (i == (bs >> 6)) jx[i] = kNoIdx;
      }
      if (lane == 0) {
        a.out_idx[row * a.out_stride + a.out_off + t] = a.col_offset + (int64_t)bi;
        a.out_val[row * a.out_stride + a.out_off + t] = val_from_key<METRIC>(bk);
      }
    }
    if (lane == 0) { a.done[pos] = 1; atomicAdd(a.defer_cnt + 256 + (blockIdx.x & 255), 1u); }
  }
  }   // parts
}

template <int METRIC>
static int launch_select_m(const SelectArgs& a, bool vec4, bool staged16, void* order_temp, size_t order_temp_bytes, hipStream_t s) {
  const int64_t grid = (a.n_rows + SEL_WAVES - 1) / SEL_WAVES;
  if ((vec4 || staged16) && a.d >= 64) {
    const bool two = a.spill_cnt != nullptr && a.two_pass && (a.row_ids == nullptr || a.is_perm);
    for (int pass = 0; pass < (two ? 2 : 1); ++pass) {
      SelectArgs b = a;
      b.pass = pass;
      if (!two) b.two_pass = 0;
      const int extra = (two && pass == 0) ? 0 : a.spill_cap;
      b.maxc = ((a.lists * a.cap + extra + 63) / 64) * 64;
      int sg = (extra == 0) ? 8 : 32;       // rows with overflow entries carry many candidates
      if (extra != 0) if (const char* e = getenv("MMF_SELECT_SG")) { const int v = atoi(e); if (v == 16 || v == 32) sg = v; }
      auto kern = a.dtype == MMF_F32 ? select_staged_kernel<METRIC, MMF_F32, 8>
                  : (a.dtype == MMF_BF16 ? select_staged_kernel<METRIC, MMF_BF16, 8> : select_staged_kernel<METRIC, MMF_F16, 8>);
      if (sg == 16) kern = a.dtype == MMF_F32 ? select_staged_kernel<METRIC, MMF_F32, 16>
                           : (a.dtype == MMF_BF16 ? select_staged_kernel<METRIC, MMF_BF16, 16> : select_staged_kernel<METRIC, MMF_F16, 16>);
      if (sg == 32) kern = a.dtype == MMF_F32 ? select_staged_kernel<METRIC, MMF_F32, 32>
                           : (a.dtype == MMF_BF16 ? select_staged_kernel<METRIC, MMF_BF16, 32> : select_staged_kernel<METRIC, MMF_F16, 32>);
      const size_t lds = sizeof(float) * SEL_WAVES * (sg * SLD + SC) + (size_t)SEL_WAVES * b.maxc * 8;
      MMF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      int64_t g = grid;
      b.order_blocks = grid;
      if (pass == 1 && a.key_in) {
        if (a.is_perm) {
          // the scan took its queries with near-duplicate rows next to each other: 32 consecutive list positions ARE rows that share
          // their candidates, whatever the smallest id of each one's set (rows of a looser cluster hold different subsets of it, and
          // the key order scatters them)
          b.group_by_pos = 1;
          b.key_sorted = nullptr; b.row_sorted = nullptr;
        } else {                                      // order the waiting rows by their smallest candidate id
          hipLaunchKernelGGL(select_keys_kernel, dim3((unsigned)((a.n_rows + 255) / 256)), dim3(256), 0, s, a);
          MMF_LAUNCH_CHECK();
          size_t tb = order_temp_bytes;
          // keys are candidate ids < m, or 0xffffffff (not waiting): the low bits(m) + 1 bits order the ids and keep the marker last
          int nb = 1;
          while (nb < 32 && (int64_t(1) << (nb - 1)) < a.m) ++nb;
          MMF_HIP(hipcub::DeviceRadixSort::SortPairs(order_temp, tb, a.key_in, const_cast<uint32_t*>(a.key_sorted), a.row_in,
                                                     const_cast<uint32_t*>(a.row_sorted), (int)a.n_rows, 0, nb, s));
        }
        g = (grid + 7) / 8 * 8;
        b.order_blocks = g;
        // grouped re-rank of 32 consecutive rows of that order on the matrix cores; what it leaves is done by the launch below
        if (a.done && (vec4 || staged16) && !getenv("MMF_SELECT_NO_GROUPS")) {
          MMF_HIP(hipMemsetAsync(a.done, 0, (size_t)a.n_rows, s));
          auto gk = a.dtype == MMF_F32 ? rerank_group_kernel<METRIC, MMF_F32>
                    : (a.dtype == MMF_BF16 ? rerank_group_kernel<METRIC, MMF_BF16> : rerank_group_kernel<METRIC, MMF_F16>);
          MMF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(GroupLds)));
          const int64_t groups = ((a.n_rows + GR_Q - 1) / GR_Q + 7) / 8 * 8;
          hipLaunchKernelGGL(gk, dim3((unsigned)groups), dim3(64 * GR_W), sizeof(GroupLds), s, b);
          MMF_LAUNCH_CHECK();
          if (g > 16384) g = 16384;                   // what is left, walked by a grid-stride loop
          // Few waiting rows (the group kernel left them all: scattered rows with a few dozen candidates each) are best served by
          // 16-candidate groups — half the LDS, twice the waves; many (near-duplicate data: the rows of groups whose union did not
          // fit) by 32.  The host does not know which case it is: both shapes are launched, the wrong one leaves at once.
          if (sg == 32) {
            SelectArgs c = b;
            c.only_if = 1;
            auto k16 = a.dtype == MMF_F32 ? select_staged_kernel<METRIC, MMF_F32, 16>
                       : (a.dtype == MMF_BF16 ? select_staged_kernel<METRIC, MMF_BF16, 16> : select_staged_kernel<METRIC, MMF_F16, 16>);
            const size_t lds16 = sizeof(float) * SEL_WAVES * (16 * SLD + SC) + (size_t)SEL_WAVES * c.maxc * 8;
            MMF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16));
            hipLaunchKernelGGL(k16, dim3((unsigned)g), dim3(64 * SEL_WAVES), lds16, s, c);
            MMF_LAUNCH_CHECK();
            b.only_if = 2;
          }
        } else {
          b.done = nullptr;
        }
      } else if (pass == 1) {
        b.key_sorted = nullptr;
        b.done = nullptr;
      }
      hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(64 * SEL_WAVES), lds, s, b);
      MMF_LAUNCH_CHECK();
    }
  } else if (vec4) hipLaunchKernelGGL((select_kernel<METRIC, true>), dim3((unsigned)grid), dim3(64 * SEL_WAVES), 0, s, a);
  else hipLaunchKernelGGL((select_kernel<METRIC, false>), dim3((unsigned)grid), dim3(64 * SEL_WAVES), 0, s, a);
  MMF_LAUNCH_CHECK();
  return MMF_OK;
}

// scratch of the ordered second pass: four u32 arrays of n (rounded to 64), the done flags, the radix sort's temporary storage
size_t select_order_bytes(int64_t n) {
  const size_t nn = ((size_t)n + 63) & ~size_t(63);
  size_t tb = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                           (int)n, 0, 32, (hipStream_t)0);
  return (5 * nn + 512) * 4 + ((tb + 255) & ~size_t(255)) + 256;
}

int launch_select(const SelectProblem& p, const CandLists& L, hipStream_t s) {
  if (p.n_rows <= 0) return MMF_OK;
  if ((int64_t)L.lists * L.cap + L.spill_cap > SEL_MAXC) {
    set_error("select: %d lists x %d entries (+ %d overflow slots) exceed the per-row capacity %d", L.lists, L.cap, L.spill_cap, SEL_MAXC);
    return MMF_E_INTERNAL;
  }
  SelectArgs a{};
  a.X = p.X; a.Y = p.Y; a.n = p.n; a.m = p.m; a.d = p.d; a.dtype = p.dtype;
  a.neg_lambda = -p.lambda; a.k = p.k; a.exclude_self = p.exclude_self;
  a.row_offset = p.row_offset; a.col_offset = p.col_offset; a.rx = p.rx; a.cy = p.cy;
  a.row_ids = p.row_ids; a.n_rows = p.n_rows; a.is_perm = 0;
  if (p.perm) {
    if (p.row_ids) { set_error("select: row_ids and perm are exclusive"); return MMF_E_INTERNAL; }
    a.row_ids = p.perm; a.is_perm = 1;
  }
  a.cand_cnt = L.cnt; a.cand_ids = L.ids; a.overflow = L.overflow; a.lists = L.lists; a.cap = L.cap;
  a.cand_keys = L.keys; a.margin = L.margin; a.slot_ulp = L.slot_ulp;
  a.spill_cnt = L.spill_cnt; a.spill_ids = L.spill_ids; a.spill_cap = L.spill_cap; a.spill_stacks = L.spill_stacks;
  a.pass = 0; a.two_pass = p.two_pass ? 1 : 0;
  void* order_temp = nullptr;
  size_t order_temp_bytes = 0;
  if (p.two_pass && p.order_scratch && p.row_ids == nullptr && L.spill_cnt) {   // (p.perm is fine: the order is one of list positions)
    uint32_t* o = static_cast<uint32_t*>(p.order_scratch);
    const size_t nn = ((size_t)p.n_rows + 63) & ~size_t(63);
    a.key_in = o; a.row_in = o + nn; a.key_sorted = o + 2 * nn; a.row_sorted = o + 3 * nn;
    a.done = reinterpret_cast<uint8_t*>(o + 4 * nn);
    a.late = a.done + nn;                            // (the block is nn words: done takes its first n bytes, late the n bytes from nn on)
    MMF_HIP(hipMemsetAsync(a.late, 0, (size_t)p.n_rows, s));
    a.defer_cnt = o + 5 * nn;
    order_temp = o + 5 * nn + 512;
    order_temp_bytes = select_order_bytes(p.n_rows) - (5 * nn + 512) * 4;
    MMF_HIP(hipMemsetAsync(a.defer_cnt, 0, 2048, s));
  }
  a.out_idx = p.out_idx; a.out_val = p.out_val;
  a.out_stride = p.out_stride > 0 ? p.out_stride : p.k; a.out_off = p.out_off;
  a.floor_key_out = p.floor_key_out; a.floor_id_out = p.floor_id_out;
  a.fail_rows = p.fail_rows; a.fail_count = p.fail_count; a.cand_total = p.cand_total;
  const bool v4 = p.dtype == MMF_F32 && (p.d % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.X) & 15) == 0) &&
                  ((reinterpret_cast<uintptr_t>(p.Y) & 15) == 0);
  // 16-bit rows take the same staged kernel with 8-byte loads
  const bool s16 = p.dtype != MMF_F32 && (p.d % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.X) & 7) == 0) &&
                   ((reinterpret_cast<uintptr_t>(p.Y) & 7) == 0);
  switch (p.metric) {
    case MMF_DOT: return launch_select_m<MMF_DOT>(a, v4, s16, order_temp, order_temp_bytes, s);
    case MMF_COSINE: return launch_select_m<MMF_COSINE>(a, v4, s16, order_temp, order_temp_bytes, s);
    case MMF_NEG_SQ_L2: return launch_select_m<MMF_NEG_SQ_L2>(a, v4, s16, order_temp, order_temp_bytes, s);
    case MMF_RBF: return launch_select_m<MMF_RBF>(a, v4, s16, order_temp, order_temp_bytes, s);
  }
  set_error("select: unsupported metric %d", p.metric);
  return MMF_E_INVALID;
}

// ------------------------------------------------------------------------------------------------
// exact top-k of a FEW rows against every column (the rows the fast path could not certify, when
// there are too few of them to feed the matrix-core rescan): rowkeys writes all m canonical keys of
// a row, rowtopk picks k of them under (key desc, id asc).
//   rowkeys: grid (column blocks of 256, rows); a wave owns 64 columns, one lane per column.  Y is
//   read in coalesced [64 columns x 32 k] tiles through LDS (stride 33: conflict-free for both the
//   row-wise fill and the lane-per-row walk); the chains run k-ascending as everywhere else.
// ------------------------------------------------------------------------------------------------
struct RowKeysArgs {
  const void* X; const void* Y; int64_t m, d; int dtype; float neg_lambda; int exclude_self;
  int64_t row_offset, col_offset; const float* rx; const float* cy; const int32_t* row_ids; int64_t n_rows;
  float* keys;   // [rows][m]
};

constexpr int RK_ROWS = 4;   // rows that share one pass over Y

template <int METRIC>
__global__ __launch_bounds__(256) void rowkeys_kernel(RowKeysArgs a) {
  __shared__ float ytile[4][64][33];
  __shared__ float xrow[RK_ROWS][32];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t g0 = (int64_t)blockIdx.y * RK_ROWS;               // first of this block's rows in row_ids
  int64_t row[RK_ROWS];
#pragma unroll
  for (int r = 0; r < RK_ROWS; ++r) row[r] = a.row_ids[(g0 + r < a.n_rows) ? (g0 + r) : g0];
  const int64_t c0 = ((int64_t)blockIdx.x * 4 + wave) * 64;      // first column of this wave
  const int64_t col = c0 + lane;
  float acc[RK_ROWS];
#pragma unroll
  for (int r = 0; r < RK_ROWS; ++r) acc[r] = 0.0f;
  for (int64_t k0 = 0; k0 < a.d; k0 += 32) {
    const int kc = (a.d - k0 < 32) ? (int)(a.d - k0) : 32;
    __syncthreads();                                               // previous chunk consumed
    if (threadIdx.x < 32 * RK_ROWS && (threadIdx.x & 31) < kc)
      xrow[threadIdx.x >> 5][threadIdx.x & 31] = ld_elem(a.X, row[threadIdx.x >> 5] * a.d + k0 + (threadIdx.x & 31), a.dtype);
    for (int r = 0; r < 64; r += 2) {                              // two tile rows per instruction, lanes along k
      const int rr = r + (lane >> 5), kl = lane & 31;
      const int64_t cr = c0 + rr;
      if (cr < a.m && kl < kc) ytile[wave][rr][kl] = ld_elem(a.Y, cr * a.d + k0 + kl, a.dtype);
    }
    __syncthreads();
    if (col < a.m) {
      for (int k = 0; k < kc; ++k) {
        const float y = ytile[wave][lane][k];
#pragma unroll
        for (int r = 0; r < RK_ROWS; ++r) acc[r] = __builtin_fmaf(xrow[r][k], y, acc[r]);
      }
    }
  }
  if (col < a.m) {
#pragma unroll
    for (int r = 0; r < RK_ROWS; ++r) {
      if (g0 + r >= a.n_rows) break;
      const bool self = a.exclude_self && (a.col_offset + col == a.row_offset + row[r]);
      float kx = key_from_dot<METRIC>(acc[r], a.rx[row[r]], a.cy[col], a.neg_lambda);
      if (self || kx != kx) kx = kNegInf;                          // NaN keys rank last, as in select
      a.keys[(g0 + r) * a.m + col] = kx;
    }
  }
}

struct RowTopkArgs {
  const float* keys; int64_t m; int k; int exclude_self; int64_t row_offset, col_offset; const int32_t* row_ids;
  int64_t* out_idx; float* out_val;
};

template <int METRIC>
__global__ __launch_bounds__(1024) void rowtopk_kernel(RowTopkArgs a) {
  __shared__ float wk[16];
  __shared__ uint32_t wi[16];
  __shared__ float pk_s;
  __shared__ uint32_t pi_s;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t row = a.row_ids[blockIdx.x];
  const float* kr = a.keys + (int64_t)blockIdx.x * a.m;
  const int64_t self = a.exclude_self ? (a.row_offset + row - a.col_offset) : -1;
  float pk = 0.0f;            // previous pick; round t takes the best entry ranked strictly after it
  uint32_t pi = kNoIdx;
  for (int t = 0; t < a.k; ++t) {
    float bk = kNegInf;
    uint32_t bi = kNoIdx;
    for (int64_t j = threadIdx.x; j < a.m; j += 1024) {
      if (j == self) continue;
      const float kj = kr[j];
      const uint32_t ij = (uint32_t)j;
      if (t > 0 && !better(pk, pi, kj, ij)) continue;              // already emitted (or the pick itself)
      if (bi == kNoIdx || better(kj, ij, bk, bi)) { bk = kj; bi = ij; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ok = __shfl_xor(bk, o);
      const uint32_t oi = (uint32_t)__shfl_xor((int)bi, o);
      if (oi != kNoIdx && (bi == kNoIdx || better(ok, oi, bk, bi))) { bk = ok; bi = oi; }
    }
    if (lane == 0) { wk[wave] = bk; wi[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float fk = wk[0];
      uint32_t fi = wi[0];
      for (int w = 1; w < 16; ++w)
        if (wi[w] != kNoIdx && (fi == kNoIdx || better(wk[w], wi[w], fk, fi))) { fk = wk[w]; fi = wi[w]; }
      pk_s = fk; pi_s = fi;
      a.out_idx[row * a.k + t] = a.col_offset + (int64_t)fi;
      a.out_val[row * a.k + t] = val_from_key<METRIC>(fk);
    }
    __syncthreads();
    pk = pk_s; pi = pi_s;
  }
}

template <int METRIC>
static int launch_rows_exact_m(const RowKeysArgs& ka, const RowTopkArgs& ta, int64_t rows, hipStream_t s) {
  hipLaunchKernelGGL((rowkeys_kernel<METRIC>), dim3((unsigned)((ka.m + 255) / 256), (unsigned)((rows + RK_ROWS - 1) / RK_ROWS)), dim3(256), 0, s, ka);
  MMF_LAUNCH_CHECK();
  hipLaunchKernelGGL((rowtopk_kernel<METRIC>), dim3((unsigned)rows), dim3(1024), 0, s, ta);
  MMF_LAUNCH_CHECK();
  return MMF_OK;
}

// p.row_ids / p.n_rows name the rows; keys: scratch of p.n_rows * p.m floats.
int launch_rows_exact(const SelectProblem& p, float* keys, hipStream_t s) {
  if (p.n_rows <= 0) return MMF_OK;
  if (!p.row_ids) { set_error("rows_exact: row list missing"); return MMF_E_INTERNAL; }
  RowKeysArgs ka{p.X, p.Y, p.m, p.d, p.dtype, -p.lambda, p.exclude_self, p.row_offset, p.col_offset, p.rx, p.cy, p.row_ids, p.n_rows, keys};
  RowTopkArgs ta{keys, p.m, p.k, p.exclude_self, p.row_offset, p.col_offset, p.row_ids, p.out_idx, p.out_val};
  switch (p.metric) {
    case MMF_DOT: return launch_rows_exact_m<MMF_DOT>(ka, ta, p.n_rows, s);
    case MMF_COSINE: return launch_rows_exact_m<MMF_COSINE>(ka, ta, p.n_rows, s);
    case MMF_NEG_SQ_L2: return launch_rows_exact_m<MMF_NEG_SQ_L2>(ka, ta, p.n_rows, s);
    case MMF_RBF: return launch_rows_exact_m<MMF_RBF>(ka, ta, p.n_rows, s);
  }
  set_error("rows_exact: unsupported metric %d", p.metric);
  return MMF_E_INVALID;
}

// ------------------------------------------------------------------------------------------------
// merge two sorted [n,k] lists; one lane per row
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool better64(float ka, int64_t ia, float kb, int64_t ib) {
  if (ka != ka) ka = kNegInf;
  if (kb != kb) kb = kNegInf;
  return (ka > kb) || (ka == kb && ia < ib);
}

__global__ void topk_merge_kernel(const int64_t* __restrict__ ia, const float* __restrict__ va,
                                  const int64_t* __restrict__ ib, const float* __restrict__ vb, int64_t n, int k,
                                  int64_t* __restrict__ io, float* __restrict__ vo) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int64_t *pa = ia + r * k, *pb = ib + r * k;
  const float *qa = va + r * k, *qb = vb + r * k;
  int a = 0, b = 0, o = 0;
  while (o < k) {
    while (a < k && pa[a] < 0) ++a;
    while (b < k && pb[b] < 0) ++b;
    const bool ta = a < k, tb = b < k;
    if (!ta && !tb) break;
    bool pick_a;
    if (ta && tb) {
      if (pa[a] == pb[b]) { ++b; continue; }
      pick_a = better64(qa[a], pa[a], qb[b], pb[b]);
    } else {
      pick_a = ta;
    }
    const int64_t id = pick_a ? pa[a] : pb[b];
    const float v = pick_a ? qa[a] : qb[b];
    if (pick_a) ++a; else ++b;
    bool dup = false;
    for (int t = 0; t < o; ++t) dup |= (io[r * k + t] == id);
    if (dup) continue;
    io[r * k + o] = id;
    vo[r * k + o] = v;
    ++o;
  }
  for (; o < k; ++o) { io[r * k + o] = -1; vo[r * k + o] = kNegInf; }
}

int launch_topk_merge(const int64_t* ia, const float* va, const int64_t* ib, const float* vb, int64_t n, int k,
                      int64_t* io, float* vo, hipStream_t s) {
  if (n <= 0) return MMF_OK;
  hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ia, va, ib, vb, n, k,
                     io, vo);
  MMF_LAUNCH_CHECK();
  return MMF_OK;
}

// ------------------------------------------------------------------------------------------------
// w_e = max(0, cos(x_i, x_j)), preprocess_hypergraph.py:414-420.  nrm = clamped norms of the rows.
// ------------------------------------------------------------------------------------------------
template <bool VEC4>
__global__ void edge_cosine_kernel(const void* __restrict__ X, int64_t d, int dtype, const float* __restrict__ nrm,
                                   const int64_t* __restrict__ ei, int64_t E, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t i = ei[e], j = ei[E + e];
  const float dot = chain_rows<VEC4>(X, i, X, j, d, dtype);
  const float c = dot / (nrm[i] * nrm[j]);
  out[e] = (c > 0.0f) ? c : 0.0f;
}

int launch_edge_cosine_impl(const void* X, int64_t d, int dtype, const float* nrm, const int64_t* ei, int64_t E,
                            float* out, hipStream_t s) {
  if (E <= 0) return MMF_OK;
  const bool v4 = dtype == MMF_F32 && (d % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
  const dim3 grid((unsigned)((E + 127) / 128));
  if (v4) hipLaunchKernelGGL(edge_cosine_kernel<true>, grid, dim3(128), 0, s, X, d, dtype, nrm, ei, E, out);
  else hipLaunchKernelGGL(edge_cosine_kernel<false>, grid, dim3(128), 0, s, X, d, dtype, nrm, ei, E, out);
  MMF_LAUNCH_CHECK();
  return MMF_OK;
}

}  // namespace mmf
