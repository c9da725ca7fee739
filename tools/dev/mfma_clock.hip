// Does the fp32-MFMA rate depend on operand data (power-limited clocks)?  Measures the shader clock inside the kernel
// (s_memtime cycles per s_memrealtime 100 MHz tick) for constant vs pseudo-random operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* clk, int iters, int mode) {
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
  float xs[8], ys[8];
  for (int u = 0; u < 8; ++u) {
    h = h * 1664525u + 1013904223u;
    xs[u] = mode == 0 ? 0.f : mode == 1 ? 0.37f : ((int)(h >> 8) - (1 << 23)) * (1.f / (1 << 23));
    h = h * 1664525u + 1013904223u;
    ys[u] = mode == 0 ? 0.f : mode == 1 ? 1.21f : ((int)(h >> 8) - (1 << 23)) * (1.f / (1 << 23));
  }
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[u], ys[u], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ys[u], xs[u], acc[1], 0, 0, 0);
    }
  }
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  float s = 0; for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}
int main(int argc, char** argv) {
  const int blocks = 512; const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  float* d; unsigned long long* c; hipMalloc(&d, blocks * 256 * 4); hipMalloc(&c, blocks * 16);
  unsigned long long hc[blocks * 2];
  for (int mode = 0; mode < 3; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, c, 100, mode); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, c, iters, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hc, c, blocks * 16, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0; for (int b = 0; b < blocks; ++b) { cyc += hc[2 * b]; wall += hc[2 * b + 1]; }
    double flop = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 2;
    printf("mode %d (%s): %.2f ms  %.1f TFLOP/s  memtime/realtime ratio %.3f (x100MHz = %.0f MHz if memtime is the shader clock); cycles per MFMA %.1f\n",
           mode, mode == 0 ? "zeros" : mode == 1 ? "const" : "random", ms, flop / ms / 1e9, cyc / wall, cyc / wall * 100,
           cyc / blocks / ((double)iters * 16 * 2));   // 2 waves per SIMD share the pipe
  }
  return 0;
}
