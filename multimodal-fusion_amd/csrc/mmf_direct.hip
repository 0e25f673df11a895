// mmf_direct.hip — the rectangular WSI x TMA similarity of compute_wsi_tma_similarity
// (build_hypergraph/preprocess_hypergraph.py:248-265):  S[i][j] = exp(-lambda * sum_k (a_ik - b_jk)^2)  in the DIRECT
// difference form of :254-256, plus the five statistics of :259-265 out of the same pass.
//
// The direct form is not a contraction (no matrix cores without changing the bits: the norm expansion cancels where the
// direct form does not), so the bound is the vector ALU: one v_sub + one v_fma per pair and k — 2 VALU instructions per
// pair-k, 64 lanes per instruction per 2 cycles per SIMD = 16 pair-k per cycle per SIMD, 3.9e13 pair-k/s at 2.4 GHz.
// Register tiling keeps it there: a workgroup owns a 128 x 128 output tile, a lane 8 x 8 outputs (64 accumulators);
// per k it reads 8 + 8 operands from LDS with four ds_read_b128 (k-major tiles, the 16 lanes that share a row group
// read one address) and issues 128 VALU instructions.  Global rows are staged through registers into the k-major
// image (a transpose: ds_write_b32, 2-way conflicts at most, which a 32-bit store does not pay for), double buffered,
// one barrier per 16 k.
//
// Statistics: every workgroup reduces sum(v - p), sum((v - p)^2) in f64 around a common pivot p, min and max over its
// valid outputs and writes one partial; a single workgroup merges the partials in index order (bit-reproducible), and
// the lower median comes from the 4-pass radix select over the stored matrix (mmf_edges.hip) — five values for one
// extra read pass... of four.  Without an output matrix the rows are recomputed panel by panel for each radix pass.
#include "mmf_dev.h"
#include "mmf_host.h"

namespace mmf {

constexpr int D_BM = 128, D_BN = 128, D_KC = 16, D_LD = 132;

struct StatPartial { double s1, s2; float mn, mx; };     // same layout as in mmf_edges.hip

template <bool VEC4>
__device__ __forceinline__ f32x4 dload4(const void* base, int64_t row, int64_t k, int64_t d, int dtype) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if constexpr (VEC4) {
    if (k < d) v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + row * d + k);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (k + i < d) v[i] = ld_elem(base, row * d + k + i, dtype);
  }
  return v;
}

constexpr int EPI_EXP = 0, EPI_EXP_STATS = 1;

// X: rows [xrow0, xrow0 + n) of the query matrix are this launch's rows; out (may be null) has leading dimension m.
template <bool VEC4, int EPI>
__global__ __launch_bounds__(256, 3) void rbf_direct_tiled_kernel(const void* __restrict__ X, int64_t n, const void* __restrict__ Y,
                                                               int64_t m, int64_t d, int dtype, float neg_lambda,
                                                               float* __restrict__ out, StatPartial* __restrict__ part,
                                                               const float* __restrict__ pivot) {
  constexpr bool STATS = (EPI == EPI_EXP_STATS);
  __shared__ __attribute__((aligned(16))) float As[2][D_KC][D_LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][D_KC][D_LD];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int64_t i0 = (int64_t)blockIdx.y * D_BM, j0 = (int64_t)blockIdx.x * D_BN;

  const int srow = tid >> 2, sk = (tid & 3) * 4;
  int64_t xr[2], yr[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    xr[i] = i0 + srow + 64 * i; if (xr[i] > n - 1) xr[i] = n - 1;
    yr[i] = j0 + srow + 64 * i; if (yr[i] > m - 1) yr[i] = m - 1;
  }
  f32x4 ra[2], rb[2];
  auto gload = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ra[i] = dload4<VEC4>(X, xr[i], k0 + sk, d, dtype);
      rb[i] = dload4<VEC4>(Y, yr[i], k0 + sk, d, dtype);
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        As[buf][sk + e][srow + 64 * i] = ra[i][e];
        Bs[buf][sk + e][srow + 64 * i] = rb[i][e];
      }
  };

  float acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.0f;

  const int nk = (int)((d + D_KC - 1) / D_KC);
  gload(0);
  swrite(0);
  __syncthreads();
  for (int s = 0; s < nk; ++s) {
    const int buf = s & 1;
    if (s + 1 < nk) gload((int64_t)(s + 1) * D_KC);
#pragma unroll 2
    for (int k = 0; k < D_KC; ++k) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(&As[buf][k][ty * 8]);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(&As[buf][k][ty * 8 + 4]);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(&Bs[buf][k][tx * 4]);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(&Bs[buf][k][64 + tx * 4]);
      // k >= d is zero on both sides: t = 0, fmaf(0, 0, acc) = acc — the chain is the canonical one, k ascending
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float ai = i < 4 ? a0[i & 3] : a1[i & 3];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float bj = j < 4 ? b0[j & 3] : b1[j & 3];
          const float t = ai - bj;
          acc[i][j] = __builtin_fmaf(t, t, acc[i][j]);
        }
      }
    }
    if (s + 1 < nk) swrite(buf ^ 1);
    __syncthreads();
  }

  // epilogue: exp, 16-byte stores (the 16 lanes of a row group cover 256 contiguous bytes twice), statistics
  double s1 = 0.0, s2 = 0.0;
  float mn = __builtin_huge_valf(), mx = -__builtin_huge_valf();
  const double p = STATS ? (double)pivot[0] : 0.0;
  const bool vst = out && ((m & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t row = i0 + ty * 8 + i;
    if (row >= n) break;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t col = j0 + 64 * h + tx * 4;
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = expf(neg_lambda * acc[i][4 * h + j]);
      if (out) {
        if (vst && col + 3 < m) *reinterpret_cast<f32x4*>(out + row * m + col) = v;
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (col + j < m) out[row * m + col + j] = v[j];
        }
      }
      if constexpr (STATS) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (col + j < m) {
            const double dx = (double)v[j] - p;
            s1 += dx; s2 = __builtin_fma(dx, dx, s2);
            mn = fminf(mn, v[j]); mx = fmaxf(mx, v[j]);
          }
      }
    }
  }
  if constexpr (STATS) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o);
      mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    __shared__ StatPartial w[4];
    if ((tid & 63) == 0) w[tid >> 6] = StatPartial{s1, s2, mn, mx};
    __syncthreads();
    if (tid == 0) {
      StatPartial r = w[0];
      for (int q = 1; q < 4; ++q) { r.s1 += w[q].s1; r.s2 += w[q].s2; r.mn = fminf(r.mn, w[q].mn); r.mx = fmaxf(r.mx, w[q].mx); }
      part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = r;
    }
  }
}

// pivot for the statistics: (about) the similarity of the first pair — any common value works, it is only a shift
__global__ __launch_bounds__(64) void rbf_direct_pivot_kernel(const void* X, const void* Y, int64_t d, int dtype, float neg_lambda,
                                                              float* pivot) {
  float acc = 0.f;
  for (int64_t k = threadIdx.x; k < d; k += 64) {
    const float t = ld_elem(X, k, dtype) - ld_elem(Y, k, dtype);
    acc = __builtin_fmaf(t, t, acc);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (threadIdx.x == 0) pivot[0] = expf(neg_lambda * acc);
}

static bool direct_vec4(const void* X, const void* Y, int64_t d, int dtype) {
  return dtype == MMF_F32 && (d % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0) &&
         ((reinterpret_cast<uintptr_t>(Y) & 15) == 0);
}

int64_t rbf_direct_blocks(int64_t n, int64_t m) { return ((n + D_BM - 1) / D_BM) * ((m + D_BN - 1) / D_BN); }

int launch_rbf_direct_pivot(const void* X, const void* Y, int64_t d, int dtype, float lambda, float* pivot, hipStream_t s) {
  hipLaunchKernelGGL(rbf_direct_pivot_kernel, dim3(1), dim3(64), 0, s, X, Y, d, dtype, -lambda, pivot);
  MMF_LAUNCH_CHECK();
  return MMF_OK;
}

// out: [n, m] or null.  part / pivot: null = no statistics; else part holds rbf_direct_blocks(n, m) entries.
int launch_rbf_direct(const void* X, int64_t n, const void* Y, int64_t m, int64_t d, int dtype, float lambda, float* out,
                      void* part, const float* pivot, hipStream_t s) {
  if (n <= 0 || m <= 0) return MMF_OK;
  const dim3 grid((unsigned)((m + D_BN - 1) / D_BN), (unsigned)((n + D_BM - 1) / D_BM));
  StatPartial* sp = reinterpret_cast<StatPartial*>(part);
  const bool v4 = direct_vec4(X, Y, d, dtype);
  if (part) {
    if (v4) hipLaunchKernelGGL((rbf_direct_tiled_kernel<true, EPI_EXP_STATS>), grid, dim3(256), 0, s, X, n, Y, m, d, dtype, -lambda, out, sp, pivot);
    else hipLaunchKernelGGL((rbf_direct_tiled_kernel<false, EPI_EXP_STATS>), grid, dim3(256), 0, s, X, n, Y, m, d, dtype, -lambda, out, sp, pivot);
  } else {
    if (v4) hipLaunchKernelGGL((rbf_direct_tiled_kernel<true, EPI_EXP>), grid, dim3(256), 0, s, X, n, Y, m, d, dtype, -lambda, out, sp, pivot);
    else hipLaunchKernelGGL((rbf_direct_tiled_kernel<false, EPI_EXP>), grid, dim3(256), 0, s, X, n, Y, m, d, dtype, -lambda, out, sp, pivot);
  }
  MMF_LAUNCH_CHECK();
  return MMF_OK;
}

}  // namespace mmf
