// Does gfx950 skip the inactive 32-lane half of a wave64 VALU instruction?  Times a dependent v_fma chain under three
// exec masks: all 64 lanes, lanes 0..31 only, every other lane (32 lanes across both halves).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(float *out, int mode, int iters) {
  const int lane = threadIdx.x & 63;
  const bool on = mode == 0 ? true : (mode == 1 ? lane < 32 : (lane & 1) == 0);
  float a = (float)lane * 1e-3f, b = 1.0001f, c = 0.5f;
  if (on) {
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int k = 0; k < 64; k++) a = __builtin_fmaf(a, b, c);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main() {
  float *d; hipMalloc(&d, 256 * 4 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2)
    for (int mode = 0; mode < 3; mode++) {
      const int threads = 256 * waves_per_simd;     // 4 SIMDs x waves_per_simd waves
      hipLaunchKernelGGL(chain, dim3(256), dim3(threads), 0, 0, d, mode, 1000);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(chain, dim3(256), dim3(threads), 0, 0, d, mode, 20000);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("waves/SIMD %d mode %d (%s): %.3f ms  -> %.2f cycles per wave-instruction at 2.4 GHz\n", waves_per_simd, mode,
             mode == 0 ? "64 lanes" : (mode == 1 ? "lanes 0-31" : "even lanes"), ms, ms * 1e-3 * 2.4e9 / (20000.0 * 64) / waves_per_simd);
    }
  return 0;
}
