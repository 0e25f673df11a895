// Probe: sustained rate of v_mfma_f32_32x32x2_f32 vs v_mfma_f32_16x16x4_f32 in a bare register loop on random
// operands (one wave per SIMD x 4 accumulator chains, whole chip), i.e. which shape the chip clocks higher on.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k32(const float* in, float* out, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(t * 8 + i) & 0xfffff]; b[i] = in[(t * 8 + 4 + i) & 0xfffff]; }
  f32x16 c[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + u) & 3], b[u], c[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
  out[t] = s;
}
__global__ __launch_bounds__(256) void k16(const float* in, float* out, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(t * 8 + i) & 0xfffff]; b[i] = in[(t * 8 + 4 + i) & 0xfffff]; }
  f32x4 c[16];
  for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) c[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i + u) & 3], b[(i >> 2) & 3], c[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += c[i][r];
  out[t] = s;
}
int main() {
  const int N = 1 << 20;
  float* h = (float*)malloc(N * 4);
  srand(3);
  for (int i = 0; i < N; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  float *din, *dout;
  (void)hipMalloc(&din, N * 4); (void)hipMalloc(&dout, 256 * 8 * 256 * 4);
  (void)hipMemcpy(din, h, N * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 200000, grid = 256 * 2;   // 2 workgroups of 4 waves per CU
  for (int rep = 0; rep < 2; ++rep) {
    float ms;
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k32, dim3(grid), dim3(256), 0, 0, din, dout, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)grid * 4 * iters * 16.0 * 4096.0;
    printf("32x32x2 : %.1f ms  %.1f TFLOP/s\n", ms, fl / ms / 1e9);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k16, dim3(grid), dim3(256), 0, 0, din, dout, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    fl = (double)grid * 4 * iters * 32.0 * 2048.0;
    printf("16x16x4 : %.1f ms  %.1f TFLOP/s\n", ms, fl / ms / 1e9);
  }
  return 0;
}
