#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = threadIdx.x * 0.001f + r;
  float x = a + threadIdx.x * 1e-6f, y = b - threadIdx.x * 1e-6f;   // non-trivial operands
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
  }
  float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(int blocks, const char* tag) {
  float* d; hipMalloc(&d, blocks * 256 * 4);
  int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, 100, 0.37f, 1.21f); hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.37f, 1.21f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)blocks * 4 * iters * 8 * NACC * 2.0 * 32 * 32 * 2;
  printf("%s blocks=%d nacc=%d: %.2f ms  %.1f TFLOP/s\n", tag, blocks, NACC, ms, flop / ms / 1e9);
  hipFree(d);
}
int main() { run<1>(256, "1wave/SIMD"); run<1>(512, "2waves/SIMD"); run<4>(256, "1wave/SIMD"); run<1>(1024, "4waves/SIMD"); run<2>(512, "2w x 2acc"); return 0; }
