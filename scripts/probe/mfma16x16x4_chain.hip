// Probe: does a chain of v_mfma_f32_16x16x4_f32 accumulate each output as the canonical k-ascending fmaf chain
// (the property v_mfma_f32_32x32x2_f32 has, include/mmf_hg.h)?  Compares bitwise with fmaf on the host.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* C, int d) {   // A [16][d], B [16][d], C [16][16] = A B^T
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < d; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * d + k0 + g], B[r * d + k0 + g], acc, 0, 0, 0);
  for (int j = 0; j < 4; ++j) C[(4 * g + j) * 16 + r] = acc[j];
}
int main() {
  const int d = 64;
  float hA[16 * d], hB[16 * d], hC[256];
  srand(7);
  for (int i = 0; i < 16 * d; ++i) { hA[i] = (float)rand() / RAND_MAX - 0.5f; hB[i] = (float)rand() / RAND_MAX - 0.5f; }
  float *dA, *dB, *dC;
  (void)hipMalloc(&dA, sizeof(hA)); (void)hipMalloc(&dB, sizeof(hB)); (void)hipMalloc(&dC, sizeof(hC));
  (void)hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, d);
  (void)hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
  int same = 0, pairwise = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      float c = 0.f;
      for (int kk = 0; kk < d; ++kk) c = fmaf(hA[i * d + kk], hB[j * d + kk], c);
      float p = 0.f;   // alternative: each instruction sums its 4 products first, then adds
      for (int k0 = 0; k0 < d; k0 += 4) { float t = 0.f; for (int u = 0; u < 4; ++u) t = fmaf(hA[i * d + k0 + u], hB[j * d + k0 + u], t); p += t; }
      same += (memcmp(&c, &hC[i * 16 + j], 4) == 0);
      pairwise += (memcmp(&p, &hC[i * 16 + j], 4) == 0);
    }
  printf("16x16x4 f32 chain: %d / 256 outputs equal the k-ascending fmaf chain bitwise (%d equal the blocked variant)\n", same, pairwise);
  return 0;
}
