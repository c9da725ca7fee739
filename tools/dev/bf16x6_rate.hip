// What would a split-operand bf16 MFMA path buy the GEMM/conv set?  (parity side: bf16_split_emulation.py)
// Register-resident rate test of one wave's 64x64 output tile (2x2 blocks of 32x32), one k16 slab per inner step:
//   mode 0: fp32 MFMA           v_mfma_f32_32x32x2_f32  x 8 per block and slab                      (what ships)
//   mode 1: bf16 x 6 terms      a = hi+mid+lo (3 truncated bf16), products of total order <= 2, v_mfma_f32_32x32x16_bf16 x 6
//                               per block and slab; A split in the loop (activations), B pre-split (packed weights)
//   mode 2: as 1, A and B both split in the loop
//   mode 3: bf16 x 3 terms      a = hi+lo, hi*hi + hi*lo + lo*hi; A split in the loop, B pre-split
// No memory traffic: an upper bound of the MFMA + conversion rate.  FLOP counted as 2*M*N*K fp32-equivalent.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk(float lo16, float hi16) {     // two truncated bf16 in one dword
  return __builtin_amdgcn_perm(__float_as_uint(hi16), __float_as_uint(lo16), 0x07060302u);
}
template <int NP> __device__ __forceinline__ void split8(const float* x, bf16x8* parts) {
  u32x4 w[NP];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float r0 = x[2 * e], r1 = x[2 * e + 1];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      float h0 = __uint_as_float(__float_as_uint(r0) & 0xffff0000u), h1 = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
      w[p][e] = pk(h0, h1);
      if (p + 1 < NP) { r0 -= h0; r1 -= h1; }
    }
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) parts[p] = __builtin_bit_cast(bf16x8, w[p]);
}

template <int MODE> __global__ void __launch_bounds__(256) k(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
  float a[2][8], b[2][8];
  for (int i = 0; i < 2; ++i) for (int u = 0; u < 8; ++u) {
    h = h * 1664525u + 1013904223u; a[i][u] = ((int)(h >> 8) - (1 << 23)) * (1.f / (1 << 23));
    h = h * 1664525u + 1013904223u; b[i][u] = ((int)(h >> 8) - (1 << 23)) * (1.f / (1 << 23));
  }
  constexpr int NP = MODE == 3 ? 2 : 3;
  bf16x8 bp[2][NP];
  for (int i = 0; i < 2; ++i) split8<NP>(b[i], bp[i]);
  for (int it = 0; it < iters; ++it) {
    // the operands of the next slab "arrive": the compiler must not hoist the conversion out of the loop
    for (int i = 0; i < 2; ++i) for (int u = 0; u < 8; ++u) { asm volatile("" : "+v"(a[i][u])); asm volatile("" : "+v"(b[i][u])); }
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[2 * i + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][u], b[j][u], acc[2 * i + j], 0, 0, 0);
    } else {
      bf16x8 ap[2][NP];
#pragma unroll
      for (int i = 0; i < 2; ++i) split8<NP>(a[i], ap[i]);
      if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i) split8<NP>(b[i], bp[i]);
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int p = 0; p < NP; ++p) asm volatile("" : "+v"(bp[i][p]));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int q = 0; q < NP; ++q)
              if (p + q <= NP - 1)      // smallest terms first would be better numerically; the rate does not care
                acc[2 * i + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][p], bp[j][q], acc[2 * i + j], 0, 0, 0);
    }
  }
  float s = 0; for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> void run(float* d, int blocks, int iters, const char* what) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100); hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)blocks * 4 * iters * 2.0 * 64 * 64 * 16;     // 4 waves per block, one 64x64x16 slab per step
  printf("mode %d (%s): %.2f ms  %.1f TFLOP/s fp32-equivalent\n", MODE, what, ms, flop / ms / 1e9);
}
int main(int argc, char** argv) {
  const int blocks = 1024; const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  float* d; hipMalloc(&d, blocks * 256 * 4);
  run<0>(d, blocks, iters, "fp32 MFMA 32x32x2");
  run<1>(d, blocks, iters, "bf16 x6, A split in loop, B pre-split");
  run<2>(d, blocks, iters, "bf16 x6, A and B split in loop");
  run<3>(d, blocks, iters, "bf16 x3, A split in loop, B pre-split");
  return 0;
}
