// mmf_prep.hip — per-row scalars in canonical arithmetic.
//
// n_i = chain(x_i, x_i) is a k-ordered fmaf chain (include/mmf_hg.h), so one lane owns one row and
// walks k in order; tiles of 64 rows x 64 k are staged through LDS so the global reads stay
// coalesced (64 consecutive elements of one row per wave instruction).
// Replaces: torch.sum(features ** 2, dim=1, keepdim=True), build_hypergraph/similarity_kernel.py:43,79.
#include "mmf_dev.h"
#include "mmf_host.h"

namespace mmf {

constexpr int PREP_WAVES = 4;

__global__ __launch_bounds__(64 * PREP_WAVES) void row_scalars_kernel(const void* __restrict__ X, int64_t n,
                                                                        int64_t d, int dtype, int metric,
                                                                        float* __restrict__ out,
                                                                        uint32_t* __restrict__ max_n) {
  __shared__ float tile[PREP_WAVES][64][65];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t row0 = ((int64_t)blockIdx.x * PREP_WAVES + wave) * 64;
  float acc = 0.0f;
  for (int64_t k0 = 0; k0 < d; k0 += 64) {
    const int64_t k = k0 + lane;
#pragma unroll 8
    for (int r = 0; r < 64; ++r) {
      const int64_t row = row0 + r;
      float v = 0.0f;
      if (row < n && k < d) v = ld_elem(X, row * d + k, dtype);
      tile[wave][r][lane] = v;
    }
    __syncthreads();
    const int kend = (d - k0 < 64) ? (int)(d - k0) : 64;
    for (int kk = 0; kk < kend; ++kk) {
      const float v = tile[wave][lane][kk];
      acc = __builtin_fmaf(v, v, acc);
    }
    __syncthreads();
  }
  const int64_t row = row0 + lane;
  if (row < n) out[row] = (metric == MMF_COSINE) ? clamped_norm(acc) : acc;
  if (max_n) {  // largest squared norm (non-negative floats order like their bits); NaN never wins
    float m = (row < n && acc == acc) ? acc : 0.0f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0 && m > 0.0f) atomicMax(max_n, __float_as_uint(m));
  }
}

// f32, d % 4 == 0, 16-byte aligned rows: the same tile walk with 16-byte global loads (4 rows x 256 B per
// wave instruction), ds_write_b128 / ds_read_b128 and a 68-float row stride: for a b128 access lane r
// touches the 16-byte slot (17 r) mod 16, so the 16 lanes of a group are conflict-free.
constexpr int PV_LD = 68;
// DT: MMF_F32 (16-byte loads) or MMF_BF16 / MMF_F16 (8-byte loads, exact upcasts)
template <int DT>
__device__ __forceinline__ f32x4 ld4_any(const void* base, int64_t elem) {
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

template <int DT>
__global__ __launch_bounds__(64 * PREP_WAVES) void row_scalars_vec_kernel(const void* __restrict__ X, int64_t n,
                                                                            int64_t d, int metric,
                                                                            float* __restrict__ out,
                                                                            uint32_t* __restrict__ max_n) {
  __shared__ __attribute__((aligned(16))) float tile[PREP_WAVES][64][PV_LD];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t row0 = ((int64_t)blockIdx.x * PREP_WAVES + wave) * 64;
  const int lr = lane >> 4;          // row within a group of 4
  const int lk = (lane & 15) * 4;    // k offset inside the 64-wide tile
  float acc = 0.0f;
  for (int64_t k0 = 0; k0 < d; k0 += 64) {
    f32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t row = row0 + 4 * i + lr;
      const int64_t k = k0 + lk;
      v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (row < n && k < d) v[i] = ld4_any<DT>(X, row * d + k);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) *reinterpret_cast<f32x4*>(&tile[wave][4 * i + lr][lk]) = v[i];
    __syncthreads();
    const int kend = (d - k0 < 64) ? (int)(d - k0) : 64;   // multiple of 4
    for (int kk = 0; kk < kend; kk += 4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(&tile[wave][lane][kk]);
      acc = __builtin_fmaf(t[0], t[0], acc);
      acc = __builtin_fmaf(t[1], t[1], acc);
      acc = __builtin_fmaf(t[2], t[2], acc);
      acc = __builtin_fmaf(t[3], t[3], acc);
    }
    __syncthreads();
  }
  const int64_t row = row0 + lane;
  if (row < n) out[row] = (metric == MMF_COSINE) ? clamped_norm(acc) : acc;
  if (max_n) {
    float m = (row < n && acc == acc) ? acc : 0.0f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0 && m > 0.0f) atomicMax(max_n, __float_as_uint(m));
  }
}

int launch_row_scalars(const void* X, int64_t n, int64_t d, int dtype, int metric, float* out, uint32_t* max_n,
                       hipStream_t s) {
  if (n <= 0) return MMF_OK;
  const int64_t rows_per_block = 64 * PREP_WAVES;
  const int64_t grid = (n + rows_per_block - 1) / rows_per_block;
  if ((d % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & (dtype == MMF_F32 ? 15 : 7)) == 0)) {
    if (dtype == MMF_F32) hipLaunchKernelGGL(row_scalars_vec_kernel<MMF_F32>, dim3((unsigned)grid), dim3(64 * PREP_WAVES), 0, s, X, n, d, metric, out, max_n);
    else if (dtype == MMF_BF16) hipLaunchKernelGGL(row_scalars_vec_kernel<MMF_BF16>, dim3((unsigned)grid), dim3(64 * PREP_WAVES), 0, s, X, n, d, metric, out, max_n);
    else hipLaunchKernelGGL(row_scalars_vec_kernel<MMF_F16>, dim3((unsigned)grid), dim3(64 * PREP_WAVES), 0, s, X, n, d, metric, out, max_n);
    MMF_LAUNCH_CHECK();
    return MMF_OK;
  }
  hipLaunchKernelGGL(row_scalars_kernel, dim3((unsigned)grid), dim3(64 * PREP_WAVES), 0, s, X, n, d, dtype,
                     metric, out, max_n);
  MMF_LAUNCH_CHECK();
  return MMF_OK;
}

// ------------------------------------------------------------------------------------------------
// The f32 operand image of the exact scan and the dense outputs (mmf_scan_f32.hip): rows converted to f32, padded with
// zero rows to a multiple of 128 and with zero columns to a multiple of 32, and DE-INTERLEAVED inside every group of
// eight k — k0 k2 k4 k6 | k1 k3 k5 k7 — so that a 16-byte unit holds what ONE lane half of v_mfma_f32_32x32x2_f32
// feeds to four consecutive k-steps.  The scan's LDS-DMA then copies units as they are and its chain reads one
// ds_read_b128 per operand per four k-steps (zero padding: fmaf(0, 0, acc) = acc, the chain's bits do not change).
// row_ids (optional): image row r is row row_ids[r] of X — the flagged rows of a rescan, gathered.
__global__ __launch_bounds__(256) void prep_f32_kernel(const void* __restrict__ X, int64_t n, int64_t d, int dtype,
                                                       const int32_t* __restrict__ row_ids, float* __restrict__ out,
                                                       int64_t rows_pad, int64_t dpad, int vec) {
  const int64_t gpr = dpad >> 3;                                   // groups of eight per image row
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= rows_pad * gpr) return;
  const int64_t r = g / gpr, k0 = (g - r * gpr) * 8;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 0.0f;
  if (r < n) {
    const int64_t src = row_ids ? (int64_t)row_ids[r] : r;
    if (vec && k0 + 8 <= d) {
      if (dtype == MMF_F32) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(X) + src * d + k0);
        const f32x4 b = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(X) + src * d + k0 + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
      } else {
        const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(X) + src * d + k0);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint16_t lo = (uint16_t)(w[i] & 0xffffu), hi = (uint16_t)(w[i] >> 16);
          if (dtype == MMF_BF16) { v[2 * i] = bf16_bits_to_f32(lo); v[2 * i + 1] = bf16_bits_to_f32(hi); }
          else {
            _Float16 hl, hh;
            __builtin_memcpy(&hl, &lo, 2); __builtin_memcpy(&hh, &hi, 2);
            v[2 * i] = (float)hl; v[2 * i + 1] = (float)hh;
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (k0 + i < d) v[i] = ld_elem(X, src * d + k0 + i, dtype);
    }
  }
  float* o = out + r * dpad + k0;
  *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[2], v[4], v[6]};
  *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[1], v[3], v[5], v[7]};
}

int64_t prep_f32_rows(int64_t n) { return (n + 127) / 128 * 128; }
int64_t prep_f32_dim(int64_t d) { return (d + 31) / 32 * 32; }
// + one block of 128 rows that is never written: a query tile of a launch whose first row is not a multiple of 128
// (panel forms) reads up to 127 rows past the image; they only reach masked outputs, but must be mapped
size_t prep_f32_bytes(int64_t n, int64_t d) { return (size_t)(prep_f32_rows(n) + 128) * (size_t)prep_f32_dim(d) * sizeof(float); }

int launch_prep_f32(const void* X, int64_t n, int64_t d, int dtype, const int32_t* row_ids, float* out, hipStream_t s) {
  if (n <= 0) return MMF_OK;
  const int64_t rows_pad = prep_f32_rows(n), dpad = prep_f32_dim(d);
  const int64_t groups = rows_pad * (dpad >> 3);
  // 32-byte (f32) / 16-byte (16-bit) loads need d % 8 == 0 and an aligned base
  const int vec = ((d & 7) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0) ? 1 : 0;
  hipLaunchKernelGGL(prep_f32_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, X, n, d, dtype, row_ids, out,
                     rows_pad, dpad, vec);
  MMF_LAUNCH_CHECK();
  return MMF_OK;
}

}  // namespace mmf
