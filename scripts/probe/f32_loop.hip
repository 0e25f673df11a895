// Probe: what the exact scan's inner loop sustains, piece by piece.  One workgroup = 4 waves, 2 workgroups per CU
// (as scan_f32_kernel), 128 x 128 macro tile per workgroup, v_mfma_f32_32x32x2_f32, operands from an LDS image that is
// written once.  Variants:
//   0  bare register loop (no LDS)                       3  ds_read_b128 image, four k-steps per read
//   1  5 ds_read_b32 per k-step (the kernel's loop)      4  variant 1 + one barrier per 16 k
//   2  ds_read_b64 image (k0 k2 | k1 k3), two k-steps    5  variant 4 + global -> register -> ds_write_b32 staging
//   6  variant 3 + barrier + ds_write_b128 staging       7  variant 1 with 2-step-ahead reads
//   9  [row][k] 16-byte units, ds_read_b128 + select      8  variant 9 + LDS-DMA staging
// Prints TFLOP/s of each (2 * 128 * 128 * k per workgroup and step).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 16, LD = KC + 1;

template <int V>
__global__ __launch_bounds__(256, 2) void loop_kernel(const float* __restrict__ in, float* __restrict__ out, int chunks, int64_t d) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Qs = reinterpret_cast<float*>(smem);            // [2][128][LD] (4-byte image) or [2][128][KC] swizzled
  float* Cs = Qs + 2 * 128 * 20;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, c = lane & 31;
  for (int i = tid; i < 2 * 2 * 128 * 20; i += 256) Qs[i] = in[(blockIdx.x * 977 + i) & 0xfffff];
  __syncthreads();
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float ra = in[tid], rb[4] = {in[tid + 256], in[tid + 512], in[tid + 768], in[tid + 1024]};
  const int srow = tid / 4, sk = 4 * (tid % 4);
  const float* gq = in + ((size_t)(blockIdx.x & 255) * 128 + srow) * d + sk;
  for (int s = 0; s < chunks; ++s) {
    const int buf = s & 1;
    f32x4 gq0, gq1, gc0, gc1;
    if constexpr (V == 5 || V == 6) {
      const float* g = gq + (size_t)(s & 31) * KC;
      gq0 = *reinterpret_cast<const f32x4*>(g); gq1 = *reinterpret_cast<const f32x4*>(g + 64 * d);
      gc0 = *reinterpret_cast<const f32x4*>(g + 128 * d); gc1 = *reinterpret_cast<const f32x4*>(g + 192 * d);
    }
    if constexpr (V == 0) {
#pragma unroll
      for (int k2 = 0; k2 < KC / 2; ++k2)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(rb[t], ra, acc[t], 0, 0, 0);
    } else if constexpr (V == 1 || V == 4 || V == 5) {
      const float* Qb = Qs + buf * 128 * LD + (32 * wave + c) * LD + half;
      const float* Cb = Cs + buf * 128 * LD + c * LD + half;
      float bn = Qb[0], avn[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) avn[t] = Cb[t * 32 * LD];
#pragma unroll
      for (int k2 = 0; k2 < KC / 2; ++k2) {
        const float b = bn;
        float av[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) av[t] = avn[t];
        if (k2 + 1 < KC / 2) {
          bn = Qb[2 * (k2 + 1)];
#pragma unroll
          for (int t = 0; t < 4; ++t) avn[t] = Cb[t * 32 * LD + 2 * (k2 + 1)];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], b, acc[t], 0, 0, 0);
        if (k2 + 1 < KC / 2) __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
    } else if constexpr (V == 7) {
      const float* Qb = Qs + buf * 128 * LD + (32 * wave + c) * LD + half;
      const float* Cb = Cs + buf * 128 * LD + c * LD + half;
      float b[KC / 2], av[KC / 2][4];
#pragma unroll
      for (int k2 = 0; k2 < KC / 2; ++k2) {
        b[k2] = Qb[2 * k2];
#pragma unroll
        for (int t = 0; t < 4; ++t) av[k2][t] = Cb[t * 32 * LD + 2 * k2];
      }
#pragma unroll
      for (int k2 = 0; k2 < KC / 2; ++k2)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k2][t], b[k2], acc[t], 0, 0, 0);
    } else if constexpr (V == 2) {
      // 8-byte image: row stride KC + 2 floats; inside a group of four k: [k0 k2 | k1 k3]; half h reads the pair at 2h
      constexpr int L2 = KC + 2;
      const float* Qb = Qs + buf * 128 * L2 + (32 * wave + c) * L2 + 2 * half;
      const float* Cb = Cs + buf * 128 * L2 + c * L2 + 2 * half;
#pragma unroll
      for (int k4 = 0; k4 < KC / 4; ++k4) {
        const f32x2 b = *reinterpret_cast<const f32x2*>(Qb + 4 * k4);
        f32x2 av[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) av[t] = *reinterpret_cast<const f32x2*>(Cb + t * 32 * L2 + 4 * k4);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t][e], b[e], acc[t], 0, 0, 0);
      }
    } else if constexpr (V == 3 || V == 6) {
      // 16-byte image: inside a group of eight k: [k0 k2 k4 k6 | k1 k3 k5 k7]; row stride KC floats, rows XOR-swizzled
      // by 16-byte units so that the 16 lanes of a ds_read_b128 group hit distinct bank quads
      const int unit = half;          // 16-byte unit inside the group of eight
      auto rd = [&](const float* base, int row, int g) {
        const int u = (2 * g + unit) ^ ((row >> 2) & 3);
        return *reinterpret_cast<const f32x4*>(base + row * KC + 4 * u);
      };
      const float* Qbuf = Qs + buf * 128 * KC;
      const float* Cbuf = Cs + buf * 128 * KC;
#pragma unroll
      for (int g = 0; g < KC / 8; ++g) {
        const f32x4 b = rd(Qbuf, 32 * wave + c, g);
        f32x4 av[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) av[t] = rd(Cbuf, 32 * t + c, g);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t][e], b[e], acc[t], 0, 0, 0);
      }
    }
    if constexpr (V == 5) {
      float* qd = Qs + (buf ^ 1) * 128 * LD + srow * LD + sk;
      float* cd = Cs + (buf ^ 1) * 128 * LD + srow * LD + sk;
#pragma unroll
      for (int e = 0; e < 4; ++e) { qd[e] = gq0[e]; qd[64 * LD + e] = gq1[e]; cd[e] = gc0[e]; cd[64 * LD + e] = gc1[e]; }
    }
    if constexpr (V == 6) {
      // a thread holds k..k+3 of a row: it is one half of two 16-byte units -> scatter as 2 x (2 floats)?  Here the
      // staging thread instead loads k {0,2,4,6} / {1,3,5,7} of its row with two 16-byte global loads in the real
      // kernel; the probe writes the registers it has as one ds_write_b128 per row (same instruction count)
      const int u0 = (sk >> 2) ^ ((srow >> 2) & 3);
      float* qd = Qs + (buf ^ 1) * 128 * KC;
      float* cd = Cs + (buf ^ 1) * 128 * KC;
      *reinterpret_cast<f32x4*>(qd + srow * KC + 4 * u0) = gq0;
      *reinterpret_cast<f32x4*>(qd + (srow + 64) * KC + 4 * (u0 ^ 0)) = gq1;
      *reinterpret_cast<f32x4*>(cd + srow * KC + 4 * u0) = gc0;
      *reinterpret_cast<f32x4*>(cd + (srow + 64) * KC + 4 * (u0 ^ 0)) = gc1;
    }
    if constexpr (V == 8 || V == 9) {
      // ---- chain: one ds_read_b128 per operand per four k (both lane halves read the same unit), per-half select ----
      auto rd = [&](const float* base, int row, int u) {
        return *reinterpret_cast<const f32x4*>(base + row * KC + 4 * (u ^ ((row >> 2) & 3)));
      };
      const float* Qbuf = Qs + buf * 128 * KC;
      const float* Cbuf = Cs + buf * 128 * KC;
      if constexpr (V == 8) {
        // LDS-DMA of the next chunk: 256 rows x 64 B = 16 pieces of 1 KiB, 4 per wave; lane -> (row, unit) of its piece
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, -1, 0x00020000);
        float* dst = Qs + (buf ^ 1) * 128 * KC;     // Q rows then (at Cs) C rows: the probe's two images are 2*128*20 floats apart
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int p = wave + 4 * i;                // piece 0..15: pieces 0..7 -> Q rows, 8..15 -> C rows
          const int row = (p & 7) * 16 + (lane >> 2);
          const int pu = lane & 3;
          const int lu = pu ^ ((row >> 2) & 3);
          const uint32_t voff = (uint32_t)((((size_t)(blockIdx.x & 255) * 128 + row + (p >> 3) * 128) * d + 4 * lu) * 4);
          float* base = (p >> 3) ? (Cs + (buf ^ 1) * 128 * KC) : dst;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(base + (p & 7) * 16 * KC), 16,
                                                   (int)voff, (int)((s & 31) * KC * 4), 0, 0);
        }
      }
#pragma unroll
      for (int u = 0; u < KC / 4; ++u) {
        const f32x4 b4 = rd(Qbuf, 32 * wave + c, u);
        f32x4 a4[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) a4[t] = rd(Cbuf, 32 * t + c, u);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float b = half ? b4[2 * e + 1] : b4[2 * e];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float av = half ? a4[t][2 * e + 1] : a4[t][2 * e];
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[t], 0, 0, 0);
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if constexpr (V == 10 || V == 11) {
      const int rq = 32 * wave + c;
      const float* Qrow = Qs + buf * 128 * KC + rq * KC;
      const float* Crow = Cs + buf * 128 * KC + c * KC;          // + t * 32 rows; (row >> 2) & 3 is the same for c and c + 32 t
      const int fq = (rq >> 2) & 3, fc = (c >> 2) & 3;
      if constexpr (V == 10) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, -1, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int p = wave + 4 * i;
          const int row = (p & 7) * 16 + (lane >> 2);
          const int lu = (lane & 3) ^ ((row >> 2) & 3);
          const uint32_t voff = (uint32_t)((((size_t)(blockIdx.x & 255) * 128 + row + (p >> 3) * 128) * d + 4 * lu) * 4);
          float* base = ((p >> 3) ? Cs : Qs) + (buf ^ 1) * 128 * KC;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(base + (p & 7) * 16 * KC), 16,
                                                   (int)voff, (int)((s & 31) * KC * 4), 0, 0);
        }
      }
      f32x4 bq_n = *reinterpret_cast<const f32x4*>(Qrow + 4 * (0 ^ fq));
      f32x4 a_n[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) a_n[t] = *reinterpret_cast<const f32x4*>(Crow + t * 32 * KC + 4 * (0 ^ fc));
#pragma unroll
      for (int u = 0; u < KC / 4; ++u) {
        const f32x4 bq = bq_n;
        f32x4 a4[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) a4[t] = a_n[t];
        if (u + 1 < KC / 4) {
          bq_n = *reinterpret_cast<const f32x4*>(Qrow + 4 * ((u + 1) ^ fq));
#pragma unroll
          for (int t = 0; t < 4; ++t) a_n[t] = *reinterpret_cast<const f32x4*>(Crow + t * 32 * KC + 4 * ((u + 1) ^ fc));
        }
        __builtin_amdgcn_sched_barrier(0);
        {
          const float b = half ? bq[1] : bq[0];
          float av[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) av[t] = half ? a4[t][1] : a4[t][0];
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], b, acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
          const float b = half ? bq[3] : bq[2];
          float av[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) av[t] = half ? a4[t][3] : a4[t][2];
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], b, acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (V == 10) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if constexpr (V == 12 || V == 13) {
      // pre-permuted global image (inside every group of eight k: k0 k2 k4 k6 | k1 k3 k5 k7): the DMA copies 16-byte
      // units as they are, the chain reads one unit per operand per four k-steps, no selects
      const int rq = 32 * wave + c;
      const float* Qrow = Qs + buf * 128 * KC + rq * KC;
      const float* Crow = Cs + buf * 128 * KC + c * KC;
      const int fq = (rq >> 2) & 3, fc = (c >> 2) & 3;
      {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, -1, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int p = wave + 4 * i;
          const int row = (p & 7) * 16 + (lane >> 2);
          const int lu = (lane & 3) ^ ((row >> 2) & 3);
          const uint32_t voff = (uint32_t)((((size_t)(blockIdx.x & 255) * 128 + row + (p >> 3) * 128) * d + 4 * lu) * 4);
          float* base = ((p >> 3) ? Cs : Qs) + (buf ^ 1) * 128 * KC;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(base + (p & 7) * 16 * KC), 16,
                                                   (int)voff, (int)((s & 31) * KC * 4), 0, 0);
        }
      }
      f32x4 bq_n = *reinterpret_cast<const f32x4*>(Qrow + 4 * ((0 + half) ^ fq));
      f32x4 a_n[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) a_n[t] = *reinterpret_cast<const f32x4*>(Crow + t * 32 * KC + 4 * ((0 + half) ^ fc));
#pragma unroll
      for (int g = 0; g < KC / 8; ++g) {
        const f32x4 bq = bq_n;
        f32x4 a4[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) a4[t] = a_n[t];
        if (g + 1 < KC / 8) {
          bq_n = *reinterpret_cast<const f32x4*>(Qrow + 4 * ((2 * (g + 1) + half) ^ fq));
#pragma unroll
          for (int t = 0; t < 4; ++t) a_n[t] = *reinterpret_cast<const f32x4*>(Crow + t * 32 * KC + 4 * ((2 * (g + 1) + half) ^ fc));
        }
        if constexpr (V == 13) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[t][e], bq[e], acc[t], 0, 0, 0);
        }
        if constexpr (V == 13) __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if constexpr (V >= 4 && V != 7) __syncthreads();
    else asm volatile("" ::: "memory");
  }
  float sum = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) sum += acc[t][r];
  out[(size_t)blockIdx.x * 256 + tid] = sum;
}

template <int V>
static void run(const float* din, float* dout, const char* name) {
  const int grid = 512, chunks = 20000;
  const size_t lds = 4 * 128 * 20 * 4;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(loop_kernel<V>, dim3(grid), dim3(256), lds, 0, din, dout, chunks, (int64_t)512);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double fl = (double)grid * chunks * 2.0 * 128 * 128 * KC;
  printf("variant %d  %-58s %8.2f ms  %6.1f TFLOP/s  %.3f of 157.3\n", V, name, best, fl / best / 1e9, fl / best / 1e9 / 157.3);
  fflush(stdout);
}

int main() {
  const int N = 1 << 20;
  float* h = (float*)malloc((size_t)N * 4 + (size_t)256 * 256 * 512 * 4);
  srand(3);
  for (int i = 0; i < N + 256 * 256 * 512; ++i) h[i] = (float)(rand() & 0xffff) / 65536.0f - 0.5f;
  float *din, *dout;
  (void)hipMalloc(&din, (size_t)N * 4 + (size_t)256 * 256 * 512 * 4); (void)hipMalloc(&dout, 512 * 256 * 4);
  (void)hipMemcpy(din, h, (size_t)N * 4 + (size_t)256 * 256 * 512 * 4, hipMemcpyHostToDevice);
  run<0>(din, dout, "bare register loop");
  run<1>(din, dout, "5 ds_read_b32 per k-step, read one step ahead");
  run<7>(din, dout, "ds_read_b32, the whole chunk's reads up front");
  run<2>(din, dout, "ds_read_b64 image, two k-steps per read");
  run<3>(din, dout, "ds_read_b128 image, four k-steps per read");
  run<4>(din, dout, "variant 1 + barrier per 16 k");
  run<5>(din, dout, "variant 4 + global->LDS staging (ds_write_b32)");
  run<6>(din, dout, "variant 3 + barrier + staging (ds_write_b128)");
  run<9>(din, dout, "ds_read_b128 [row][k] image + per-half select + barrier");
  run<8>(din, dout, "variant 9 + LDS-DMA staging (16 B per lane)");
  run<11>(din, dout, "variant 9 read one unit ahead (pinned schedule)");
  run<10>(din, dout, "variant 11 + LDS-DMA staging");
  run<12>(din, dout, "pre-permuted global image: LDS-DMA + b128 reads, no selects");
  run<13>(din, dout, "variant 12, schedule pinned (reads of the next group first)");
  return 0;
}
