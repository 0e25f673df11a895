// Stand-in for a collective's kernel: `wgs` workgroups of 256 threads that stay resident for `ms` milliseconds
// (RCCL's all-gather keeps a few dozen workgroups resident for the length of the transfer).  Measurement aid
// for scripts/contention.py; not part of libmmf_hg.so.
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ __launch_bounds__(256) void occupy_kernel(uint64_t ticks, float* sink) {
  const uint64_t t0 = wall_clock64();
  float x = threadIdx.x;
  asm volatile("" ::: "v95");   // a collective kernel's register footprint (96 VGPRs), so that it cannot slip in beside full scan waves
  while (wall_clock64() - t0 < ticks) {
    for (int i = 0; i < 64; ++i) x = x * 1.0001f + 0.5f;
    __builtin_amdgcn_s_sleep(8);
  }
  if (x == 12345.678f) sink[0] = x;
}
extern "C" int occupy_launch(int wgs, double ms, void* sink, void* stream) {
  int rate_khz = 100000;
  (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
  const uint64_t ticks = (uint64_t)(ms * rate_khz);
  hipLaunchKernelGGL(occupy_kernel, dim3(wgs), dim3(256), 0, static_cast<hipStream_t>(stream), ticks, static_cast<float*>(sink));
  return (int)hipGetLastError();
}
