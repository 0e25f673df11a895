// mmf_dev.h — device-side building blocks shared by the gfx950 kernels.
//
//  * canonical f32 arithmetic (include/mmf_hg.h): the whole library is compiled with
//    -ffp-contract=off, every fused multiply-add below is an explicit fmaf;
//  * lane-private candidate lists in LDS with a wave-wide compaction: the top-k machinery both
//    scan kernels share.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MMF_DOT 0
#define MMF_COSINE 1
#define MMF_NEG_SQ_L2 2
#define MMF_RBF 3
#define MMF_RBF_DIRECT 4

namespace mmf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr float kNegInf = -__builtin_huge_valf();
constexpr float kFltMax = 3.402823466e+38f;
constexpr uint32_t kNoIdx = 0xffffffffu;

// ---------------------------------------------------------------------------------------------
// canonical arithmetic
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sq_from(float ni, float nj, float dot) {
  float s = ni + nj;      // similarity_kernel.py:49  (n_i + n_j) ...
  float t = 2.0f * dot;   // exact
  return s - t;           //                           ... - 2*dot
}

// max(sqrtf(n), 1e-8f): the clamped norm of F.cosine_similarity (preprocess_hypergraph.py:419)
__device__ __forceinline__ float clamped_norm(float n) {
  float a = __builtin_sqrtf(n);
  return (a > 1e-8f) ? a : 1e-8f;
}

// Ranking key of one pair from its canonical dot.  ri / cj are the per-row / per-column scalars
// prepared once per call: the canonical squared norm n, or clamped_norm(n) for MMF_COSINE.
template <int METRIC>
__device__ __forceinline__ float key_from_dot(float dot, float ri, float cj, float neg_lambda) {
  if constexpr (METRIC == MMF_DOT) return dot;
  if constexpr (METRIC == MMF_COSINE) return dot / (ri * cj);
  if constexpr (METRIC == MMF_NEG_SQ_L2) return -sq_from(ri, cj, dot);
  return neg_lambda * sq_from(ri, cj, dot);  // MMF_RBF: the exponent
}

template <int METRIC>
__device__ __forceinline__ float val_from_key(float key) {
  if constexpr (METRIC == MMF_RBF) return expf(key);
  return key;
}

// element loads with exact upcast to f32
__device__ __forceinline__ float ld_f32(const float* p, int64_t i) { return p[i]; }
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ float ld_elem(const void* p, int64_t i, int dtype) {
  if (dtype == 0) return ((const float*)p)[i];
  if (dtype == 1) return bf16_bits_to_f32(((const uint16_t*)p)[i]);
  return (float)(((const _Float16*)p)[i]);
}

// (key desc, id asc); used for every in-kernel ordering decision
__device__ __forceinline__ bool better(float ka, uint32_t ia, float kb, uint32_t ib) {
  return (ka > kb) || (ka == kb && ia < ib);
}

// ---------------------------------------------------------------------------------------------
// LaneList — the lane-private candidate list of the EXACT scan (mmf_scan_f32.hip): final keys.
//
// A lane owns ONE query (MFMA column = lane & 31) and sees half of every 32-candidate sub-tile
// (lane >> 5 picks the half), so candidates of a query are collected by exactly two lanes, l and
// l ^ 32, each into its own list: no atomics, no cross-wave traffic.  Entry e of thread t lives at
// keys[e * NT + t] / ids[e * NT + t] (NT = threads per block): consecutive lanes hit consecutive banks.
//
// Invariant: thr <= kk-th best key of the query over everything scanned so far, so a column that can
// still belong to the final top-kk always passes `key >= thr`.  Keys are final, so a compaction
// truncates the pair of lists to the top kk by the total order (key desc, id asc) and nothing is ever
// lost: the list cannot overflow (`overflow` is a guard that stays 0).
// ---------------------------------------------------------------------------------------------
template <int CAP, int NT>
struct LaneList {
  float* keys;     // LDS, already offset by threadIdx.x
  uint32_t* ids;   // LDS, already offset by threadIdx.x
  int cnt;
  float thr;
  uint32_t overflow;

  __device__ __forceinline__ void init(float* k, uint32_t* i) {
    keys = k; ids = i; cnt = 0; thr = -kFltMax; overflow = 0;
  }

  __device__ __forceinline__ void push(float key, uint32_t id) {
    keys[cnt * NT] = key;
    ids[cnt * NT] = id;
    ++cnt;
  }

  // The total order (key desc, id asc) as ONE signed 64-bit compare of (order-preserving key bits : ~id);
  // -0.0 is folded into +0.0 first, as the float compare of better() does.
  static __device__ __forceinline__ long long order64(float key, uint32_t id) {
    const int b = __float_as_int(key + 0.0f);
    const int e = b >= 0 ? b : (b ^ 0x7fffffff);
    return (long long)(((unsigned long long)(uint32_t)e << 32) | (unsigned long long)(~id));
  }

  // Wave-wide (every lane of the wave calls it, EXEC full).  kk = entries a query must retain.  Every entry is
  // ranked by COUNTING the entries of its own and its partner lane's list that beat it, straight out of LDS (same
  // wave, so the partner's writes are ordered before these reads), eight entries per sweep: one pair of ds_reads
  // feeds eight compares and the reads of a sweep pipeline (one entry per sweep waited for every read; a (key, id)
  // sorting network in registers, the first version for 16 entries, cost the hot loop 1 400 spilled scalar registers).
  // The entry of union rank kk - 1 is the new threshold; a lane keeps what passes it among its own top kk.
  __device__ __forceinline__ void compact(int kk) {
    constexpr int BLK = 8;
    const int pofs = (int)((threadIdx.x ^ 32u) - threadIdx.x);
    const int pcnt = __shfl_xor(cnt, 32);
    unsigned long long topmask = 0;
    float t_own = kNegInf;
    for (int e0 = 0; e0 < cnt; e0 += BLK) {
      float kf32[BLK]; long long ke[BLK]; int ro[BLK], rp[BLK];
#pragma unroll
      for (int i = 0; i < BLK; ++i) {                   // rows beyond cnt: stale but inside the lane's column; masked below
        const int e = (e0 + i < CAP) ? e0 + i : CAP - 1;
        kf32[i] = keys[e * NT]; ke[i] = order64(kf32[i], ids[e * NT]); ro[i] = 0; rp[i] = 0;
      }
      for (int f = 0; f < cnt; ++f) {
        const long long kf = order64(keys[f * NT], ids[f * NT]);
#pragma unroll
        for (int i = 0; i < BLK; ++i) ro[i] += (kf > ke[i]) ? 1 : 0;
      }
      for (int f = 0; f < pcnt; ++f) {
        const long long kf = order64(keys[f * NT + pofs], ids[f * NT + pofs]);
#pragma unroll
        for (int i = 0; i < BLK; ++i) rp[i] += (kf > ke[i]) ? 1 : 0;
      }
#pragma unroll
      for (int i = 0; i < BLK; ++i) {
        const bool live = e0 + i < cnt;
        if (live && ro[i] + rp[i] == kk - 1) t_own = kf32[i];
        if (live && ro[i] < kk) topmask |= 1ull << (e0 + i);
      }
    }
    const float t = fmaxf(t_own, __shfl_xor(t_own, 32));
    if (t != kNegInf && t > thr) thr = t;               // fewer than kk entries in the union: keep collecting everything
    int w = 0;
    for (int e = 0; e < cnt; ++e) {
      const float ke = keys[e * NT];
      const uint32_t ie = ids[e * NT];
      if ((ke >= thr) && ((topmask >> e) & 1ull)) {
        keys[w * NT] = ke;
        ids[w * NT] = ie;
        ++w;
      }
    }
    cnt = w;
  }

  // Offer the 16 keys one lane holds for one 32x32 accumulator tile.  v[r] belongs to candidate
  // id0 + (r&3) + 8*(r>>2) + 4*(lane>>5).  Called by the whole wave when any lane has a hit.
  __device__ __forceinline__ void offer_tile(const f32x16& v, uint32_t id0, int half, int kk) {
    // only the accumulator rows that hold a hit in some lane run the push code (16 branch-free compares find them;
    // thr only rises meanwhile, so the mask is a superset)
    uint32_t rmask = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) rmask |= (__any(v[r] >= thr) ? 1u : 0u) << r;
    while (rmask) {
      const int r = __builtin_ctz(rmask);
      rmask &= rmask - 1;
      const float x = v[r];
      bool hit = x >= thr;
      if (__any(hit && cnt >= CAP)) {
        compact(kk);
        hit = x >= thr;
      }
      if (hit) {
        if (cnt < CAP) push(x, id0 + (uint32_t)((r & 3) + 8 * (r >> 2) + 4 * half));
        else overflow = 1;
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------
// SlotList — the candidate list of the 16-bit scan (approximate keys only).
//
// Keys live in LDS (entry e of thread t at keys[e * NT + t]); the low 4 mantissa bits of a stored
// key carry a SLOT number, and the candidate's column id sits in a lane-private GLOBAL slot array
// (idslot[s * NT + t]) that is written once on push and read once when the scan ends.  Compaction
// therefore sorts / filters bare keys in LDS and never touches global memory.  The <= 15 ulp the
// slot bits perturb a key by are part of the error margin (scan kernel, E1).
// A lane owns one query and half of every tile like LaneList (partner lane l ^ 32); keys are APPROXIMATE here, so
//   * the threshold is  thr = (kk-th best key of the pair's union) - margin  (margin = 2 x the proven error bound of
//     an approximate key, scan kernel): every column whose EXACT key can be in the final top-kk passes `key >= thr`;
//   * a list can be crowded — more entries inside the band [t - margin, t] than it holds (near-duplicate data).  What
//     it cannot hold goes to the row's overflow list (SpillSink); without one, or with it full, the best dropped key
//     is remembered in `lost`, the threshold rises to it, and after the last launch the audit flags the row for an
//     exact rescan if that key reaches the best threshold any list of the row proved (audited loss: a transient
//     crowd early in a scan, far below the final band, costs nothing).
// ---------------------------------------------------------------------------------------------
// Where a full list puts what it cannot hold: the row's overflow list in global memory (one counter + `cap` id
// slots per query, shared by every lane, workgroup and launch that scans the query).  Near-duplicate data puts far
// more columns inside a query's margin band than a list has entries; they are kept here, not dropped, and the
// exact re-rank reads them with the lists.  cap == 0: no sink (a full list then loses the key, audited).
#ifndef MMF_SINK_CHUNK
#define MMF_SINK_CHUNK 8u      // A/B switches of scripts/ab_build.sh: -DMMF_SINK_CHUNK=1u, -DMMF_BAND_DIRECT=0
#endif
// A crowded list sends a hit to the overflow list not only below the proven k-th best key (`tband`) but up to one margin above it:
// pushing such a hit could raise the threshold by less than that margin, while on near-duplicate data (a cluster's cosines differ
// by 5e-5 .. 2e-3, the margin is 1e-3) every later member of the cluster lies just above `tband` and each push ends in a wave-wide
// compaction sooner or later.  The band a row writes down becomes at most twice as wide (measured: 129 -> 129 candidates per row on
// the bench's clusters, 94 -> 106 on clusters with cosines 0.99; a quarter margin: 57.6 -> 53.6 ms / 60.5 -> 59.2 ms of scan, one
// margin: 53.6 / 57.5 ms, four margins: the same).  -DMMF_BAND_SLACK=0.0f: A/B.
#ifndef MMF_BAND_SLACK
#define MMF_BAND_SLACK 1.0f
#endif
// A compaction reads column ids back from global memory.  Left pending, those loads make the compiler open the row loop of offer_tile
// with s_waitcnt vmcnt(0) — on EVERY iteration, i.e. every overflow-list store of the previous row (stores count in vmcnt on gfx9) is
// waited for before the next row starts.  Draining inside the rare compaction branch leaves the loop header without a wait (ISA checked):
// clustered rows -0.9 %, N = 65536 -2.4 %, the headline -0.45 % (same-process A/B).  '-DMMF_DRAIN_AFTER_COMPACT=(void)0': A/B.
#ifndef MMF_DRAIN_AFTER_COMPACT
#define MMF_DRAIN_AFTER_COMPACT __builtin_amdgcn_s_waitcnt(0x0F70)
#endif
#ifndef MMF_BAND_DIRECT
#define MMF_BAND_DIRECT 1
#endif
struct SpillSink {
  uint32_t* cnt = nullptr;    // [rows]
  uint32_t* ids = nullptr;    // [rows][cap]
  uint32_t cap = 0;
  uint32_t seg_len = 0, seg_stride = 0, id_off = 0;   // operand column -> reported id (ScanB16Args)
  int64_t row = 0;
  int ablate = 0;   // timing-only (MMF_SCAN_DEBUG bit 128): pretend the entry was stored
  // A lane reserves slots with a returning atomic on the row's counter — a round trip to the L2 the wave waits for.  A
  // lane that is known to have many entries coming (bulk: its list is crowded) reserves kChunk slots at a time and fills
  // them privately; what it has not used when the scan ends is closed with kEmpty, which the reader skips.  The
  // counter therefore counts RESERVED slots and may pass `cap` (the reader clamps; a reservation that starts at or
  // beyond cap fails, and the caller records the key as lost).
  static constexpr uint32_t kChunk = MMF_SINK_CHUNK, kEmpty = 0xffffffffu;
  uint32_t cpos = 0, cleft = 0;
  // stacks != 0 — the row has ONE pair of lists (one launch, no column splits: exactly two lanes ever store for it): no counter
  // at all.  The lower lane (stacks = 1) fills the row's slots from the front, the upper one (stacks = 2) from the back; each
  // counts its own entries (cpos) and the pair writes both counts when the scan ends (low / high 16 bits of cnt[row]).  If the
  // two stacks met — more than `cap` entries — the row is flagged for the exact rescan, as when the counter passed cap.  The
  // returning atomics of the counter form (a round trip to the L2 the wave sits out once per eight entries) were 10 of the
  // 72 ms of the scan on the clustered benchmark.
  int stacks = 0;
  __device__ __forceinline__ bool put(uint32_t id, bool bulk) {
    if (cap == 0) return false;
    if (ablate) return true;
    if (stacks) {
      if (cpos >= cap) return false;
      if (seg_len) id = (id / seg_len) * seg_stride + id % seg_len;
      ids[row * cap + (stacks == 1 ? cpos : cap - 1u - cpos)] = id + id_off;
      ++cpos;
      return true;
    }
    if (cleft == 0) {
      const uint32_t want = bulk ? kChunk : 1u;
      const uint32_t base = atomicAdd(cnt + row, want);
      if (base >= cap) return false;
      cpos = base;
      cleft = (cap - base < want) ? (cap - base) : want;
    }
    if (seg_len) id = (id / seg_len) * seg_stride + id % seg_len;
    ids[row * cap + cpos] = id + id_off;
    ++cpos; --cleft;
    return true;
  }
  __device__ __forceinline__ void close() {
    if (stacks) return;
    while (cleft != 0) { ids[row * cap + cpos] = kEmpty; ++cpos; --cleft; }
  }
};

template <int CAP, int NT>
struct SlotList {
  static_assert(CAP <= 32, "at most 5 slot bits");
  static constexpr int SLOTBITS = (CAP <= 16) ? 4 : 5;          // low mantissa bits of a stored key that carry its id slot
  static constexpr uint32_t SLOTMASK = (1u << SLOTBITS) - 1u;
  static constexpr int SLOTS = 1 << SLOTBITS;                  // id slots per lane in the global slot block
  SpillSink sink;
  float* keys;        // LDS, offset by threadIdx.x
  uint32_t* idslot;   // global, offset by threadIdx.x
  int cnt;
  uint32_t used;      // bit s set: slot s holds the id of a live entry
  float thr;
  float lost;
  uint32_t overflow;
  float tband;        // crowded lists only: the k-th best key the partner lists have proved (top of the margin band), else +inf..-inf
                      //   = -inf until a compaction finds the band fuller than the list: from then on a hit BELOW it cannot move
                      //   the threshold any more and goes straight to the overflow list (no push, no compaction)

  __device__ __forceinline__ void init(float* k, uint32_t* ids) {
    keys = k; idslot = ids; cnt = 0; used = 0; thr = -kFltMax; lost = kNegInf; overflow = 0; tband = kNegInf;
  }
  __device__ __forceinline__ void finish() {
    if (lost >= thr) overflow = 1;
    sink.close();
  }
  __device__ __forceinline__ void push(float key, uint32_t id) {   // requires cnt < CAP
    const int sl = __builtin_ctz(~used);
    idslot[sl * NT] = id;
    keys[cnt * NT] = __uint_as_float((__float_as_uint(key) & ~SLOTMASK) | (uint32_t)sl);
    used |= 1u << sl;
    ++cnt;
  }
  __device__ __forceinline__ uint32_t id_of(int e) const {          // entry e -> column id (global read)
    return idslot[(__float_as_uint(keys[e * NT]) & SLOTMASK) * NT];
  }
  __device__ __forceinline__ void raise_thr(float t, float margin) {
    const float nthr = (t == kNegInf) ? -kFltMax : (t - margin);
    if (nthr > thr) thr = nthr;
  }

  // wave-wide; leaves at least two free slots (what it has to give up for that goes to the sink, or — without
  // one, or with the sink full — is recorded in `lost`)
  __device__ __forceinline__ void compact(int kk, float margin) {
    if constexpr (CAP > 16) { compact_ranked(kk, margin); return; }
    float k[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) k[e] = ((e < CAP) && (e < cnt)) ? keys[(e < CAP ? e : 0) * NT] : kNegInf;
#pragma unroll
    for (int size = 2; size <= 16; size <<= 1) {
#pragma unroll
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int j = i ^ stride;
          if (j > i) {
            const float hi = fmaxf(k[i], k[j]), lo = fminf(k[i], k[j]);
            if ((i & size) == 0) { k[i] = hi; k[j] = lo; } else { k[i] = lo; k[j] = hi; }
          }
        }
      }
    }
    // k (this lane, descending) and the partner lane's list reversed form a bitonic sequence of 32: the element-wise
    // maxima are the 16 largest of the union, the minima the 16 smallest; a 4-stage clean sorts either half.
    float pr[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) pr[e] = __shfl_xor(k[15 - e], 32);
    float c[16];
    const bool low = (CAP >= 15) && (kk > 16);     // k + self in 17..20: the threshold sits in the lower half
#pragma unroll
    for (int e = 0; e < 16; ++e) c[e] = low ? fminf(k[e], pr[e]) : fmaxf(k[e], pr[e]);
#pragma unroll
    for (int stride = 8; stride > 0; stride >>= 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int j = i ^ stride;
        if (j > i) {
          const float hi = fmaxf(c[i], c[j]), lo = fminf(c[i], c[j]);
          c[i] = hi; c[j] = lo;
        }
      }
    }
    const int want = low ? (kk - 17) : (kk - 1);
    float t = kNegInf;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      if (e == want) t = c[e];
    }
    raise_thr(t, margin);
    int keep = 0;
#pragma unroll
    for (int e = 0; e < CAP; ++e) keep += ((e < cnt) && (k[e] >= thr)) ? 1 : 0;
    float cut = thr;
    int room = CAP - 2;
    if (keep >= CAP - 1) {
      // crowded (near-duplicate columns: more of them inside the band than the list holds).  With an overflow list and
      // kk <= CAP / 2 - 1 only this lane's CAP / 2 + 1 best stay — more than the threshold (k-th best of the two
      // partner lists) can ever need of it — so the list does not fill up again at the very next hit; keeping CAP - 2
      // made every slow-path entry of a crowded wave a compaction.  The rest of the band moves to the overflow list.
      constexpr int SMALL = CAP / 2 + 1;
      const bool deep = sink.cap != 0 && kk + 2 <= SMALL;
      cut = deep ? k[SMALL - 1] : k[CAP - 3];
      room = deep ? SMALL : CAP - 2;
      if (sink.cap != 0 && t > tband) tband = t;
    }
    int w = 0;
    uint32_t nused = 0;
    for (int e = 0; e < cnt; ++e) {     // in-place filter of the LDS keys, lane-private
      const float ke = keys[e * NT];
      if (ke >= cut && w < room) {
        keys[w * NT] = ke; nused |= 1u << (__float_as_uint(ke) & SLOTMASK); ++w;
      } else if (ke >= thr) {           // still inside the band: to the overflow list, else lost (audited)
        if (!sink.put(idslot[(__float_as_uint(ke) & SLOTMASK) * NT], tband != kNegInf)) lost = fmaxf(lost, ke);
      }
    }
    thr = fmaxf(thr, lost);
    used = nused;
    cnt = w;
  }

  // CAP > 16 (k + self in 21..44): no sorting network — a lane's 32 keys do not fit registers beside the query
  // fragments.  Every entry is ranked by counting straight out of LDS, against its own list and the partner lane's
  // (same wave: its LDS writes are ordered before these reads), eight entries per sweep so that one ds_read feeds eight
  // compares and the reads of a sweep pipeline.  Keys of one list differ in their slot bits; a key equal to one of
  // the partner list's ranks behind it in the upper lane half only, so exactly one entry of the union has rank
  // kk - 1.  The union the threshold sees is what the two lists HOLD: a key that has moved to the overflow list no
  // longer counts, which can only make the threshold lower (still valid).
  static __device__ __forceinline__ int key_enc(float x) {          // order-preserving float -> int
    const int b = __float_as_int(x);
    return b >= 0 ? b : (b ^ 0x7fffffff);
  }
  __device__ __forceinline__ void compact_ranked(int kk, float margin) {
    constexpr int BLK = 8;
    const int pofs = (int)((threadIdx.x ^ 32u) - threadIdx.x);
    const int pcnt = __shfl_xor(cnt, 32);
    const int tie = (threadIdx.x & 32u) ? 1 : 0;       // upper lane half: an equal partner key ranks first
    const int room_crowded = (kk + 2 < CAP - 2) ? (kk + 2) : (CAP - 2);
    uint32_t own_top = 0;                               // entries whose rank inside THIS list is below room_crowded
    float t_own = kNegInf;
    for (int e0 = 0; e0 < cnt; e0 += BLK) {
      float kf32[BLK]; int ke[BLK], ro[BLK], rp[BLK];
#pragma unroll
      for (int i = 0; i < BLK; ++i) {                   // e0 + i < CAP: inside the lane's column (stale beyond cnt, masked below)
        kf32[i] = keys[(e0 + i) * NT]; ke[i] = key_enc(kf32[i]); ro[i] = 0; rp[i] = 0;
      }
      for (int f = 0; f < cnt; ++f) {
        const int kf = key_enc(keys[f * NT]);
#pragma unroll
        for (int i = 0; i < BLK; ++i) ro[i] += (kf > ke[i]) ? 1 : 0;
      }
      for (int f = 0; f < pcnt; ++f) {
        const int kf = key_enc(keys[f * NT + pofs]) + tie;
#pragma unroll
        for (int i = 0; i < BLK; ++i) rp[i] += (kf > ke[i]) ? 1 : 0;
      }
#pragma unroll
      for (int i = 0; i < BLK; ++i) {
        const bool live = e0 + i < cnt;
        if (live && ro[i] + rp[i] == kk - 1) t_own = kf32[i];
        if (live && ro[i] < room_crowded) own_top |= 1u << (e0 + i);
      }
    }
    const float t = fmaxf(t_own, __shfl_xor(t_own, 32));
    raise_thr(t, margin);
    int keep = 0;
    for (int e0 = 0; e0 < cnt; e0 += BLK) {
#pragma unroll
      for (int i = 0; i < BLK; ++i) keep += ((e0 + i < cnt) && keys[(e0 + i) * NT] >= thr) ? 1 : 0;
    }
    const bool crowded = keep >= CAP - 1;               // only the lane's best room_crowded entries stay then
    if (crowded && sink.cap != 0 && t > tband) tband = t;
    int w = 0;
    uint32_t nused = 0;
    for (int e0 = 0; e0 < cnt; e0 += BLK) {             // in-place filter: a block is in registers before it is overwritten
      float kb[BLK];
#pragma unroll
      for (int i = 0; i < BLK; ++i) kb[i] = keys[(e0 + i) * NT];
#pragma unroll
      for (int i = 0; i < BLK; ++i) {
        const float ke = kb[i];
        const bool live = (e0 + i < cnt) && (ke >= thr);
        const bool stay = live && (!crowded || ((own_top >> (e0 + i)) & 1u));
        if (stay && w < CAP - 2) {
          keys[w * NT] = ke; nused |= 1u << (__float_as_uint(ke) & SLOTMASK); ++w;
        } else if (live) {
          if (!sink.put(idslot[(__float_as_uint(ke) & SLOTMASK) * NT], tband != kNegInf)) lost = fmaxf(lost, ke);
        }
      }
    }
    thr = fmaxf(thr, lost);
    used = nused;
    cnt = w;
  }

  // One tile's 16 values of this lane's query.  rowof(r): tile row (0..31) of element r — the MFMA shape's C layout.
  // A pre-emptive compaction when some list is nearly full, then only the accumulator rows that hold a hit in
  // ANY lane run the push code (16 branch-free compares find them; thr only rises meanwhile, so the mask is a
  // superset).  A hit that finds its list full compacts first (wave-wide; the loop is wave-uniform) and is
  // dropped — under the audited-loss rule — only if that frees nothing.
  template <class RowOf>
  __device__ __forceinline__ void offer_tile(const f32x16& v, uint32_t id0, RowOf rowof, int kk, float margin) {
    if (__any(cnt >= CAP - 1)) { compact(kk, margin); MMF_DRAIN_AFTER_COMPACT; }
    uint32_t rmask = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) rmask |= (__any(v[r] >= thr) ? 1u : 0u) << r;
    while (rmask) {
      const int r = __builtin_ctz(rmask);
      rmask &= rmask - 1;
      const float x = v[r];
      bool hit = x >= thr;
      const bool band = MMF_BAND_DIRECT && x < tband + MMF_BAND_SLACK * margin;
      if (__any(hit && !band && cnt >= CAP)) {
        compact(kk, margin);
        MMF_DRAIN_AFTER_COMPACT;
        hit = x >= thr;
      }
      if (hit) {
        if (__builtin_expect(!band && cnt < CAP, 1)) {
          push(x, id0 + rowof(r));
        } else if (!sink.put(id0 + rowof(r), tband != kNegInf)) {     // a band entry of a crowded list, or a hit that found the list full
          lost = fmaxf(lost, x);
          thr = fmaxf(thr, lost);
        }
      }
    }
  }
};

__device__ __forceinline__ float max16(const f32x16& v) {
  float a = fmaxf(fmaxf(v[0], v[1]), v[2]);
  float b = fmaxf(fmaxf(v[3], v[4]), v[5]);
  float c = fmaxf(fmaxf(v[6], v[7]), v[8]);
  float d = fmaxf(fmaxf(v[9], v[10]), v[11]);
  float e = fmaxf(fmaxf(v[12], v[13]), v[14]);
  return fmaxf(fmaxf(fmaxf(a, b), fmaxf(c, d)), fmaxf(e, v[15]));
}

}  // namespace mmf
