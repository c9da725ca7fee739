// What does an instruction cost beside bf16 MFMAs on gfx950?  Twin of mfma_filler.hip for v_mfma_f32_32x32x16_bf16 (8 passes =
// 32 cycles): each wave issues chains of MFMAs (on ONE accumulator: dependent, or alternating between two) and after every
// MFMA N fillers of one kind.  Reported: ns per MFMA per SIMD (32 cycles = 13.3 ns at 2.4 GHz), cycles per filler.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
enum { K_NONE, K_FMA, K_EXP, K_AND, K_PERM, K_SUB };
template <int KIND, int N, int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  f32x16 acc[2];
  for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x); b[i] = (short)0x3f80; }
  float fa = threadIdx.x * 1e-3f, fb = 1.0001f, x[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
  unsigned u[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      acc[s % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[s % NACC], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < N; ++f) {
        if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[f & 7]) : "v"(fb), "v"(fa));
        if (KIND == K_SUB) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[f & 7]) : "v"(fa));
        if (KIND == K_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[f & 7]));
        if (KIND == K_AND) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[f & 7]));
        if (KIND == K_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[f & 7]) : "v"(u[(f + 1) & 7]), "s"(0x07060302));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int f = 0; f < 8; ++f) s += x[f] + u[f];
  for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
static double base_ns;
template <int KIND, int N, int NACC> void run(float* d, int occ, int iters, const char* what) {
  const int blocks = 256 * occ;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, N, NACC>), dim3(blocks), dim3(256), 0, 0, d, 10); hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, N, NACC>), dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ns = ms * 1e6 / ((double)occ * iters * 16);
  if (KIND == K_NONE) base_ns = ns;
  printf("%d waves/SIMD %d acc  %-22s %6.1f ns per MFMA per SIMD", occ, NACC, what, ns);
  if (KIND != K_NONE) printf("   = %+.1f cycles per filler (bare chain = 32 cycles)", (ns - base_ns) / base_ns * 32 / N);
  printf("\n");
}
template <int NACC> void sweep(float* d, int occ, int iters) {
  run<K_NONE, 0, NACC>(d, occ, iters, "bare MFMA chain");
  run<K_FMA, 2, NACC>(d, occ, iters, "2 x v_fma_f32");
  run<K_FMA, 4, NACC>(d, occ, iters, "4 x v_fma_f32");
  run<K_FMA, 8, NACC>(d, occ, iters, "8 x v_fma_f32");
  run<K_FMA, 16, NACC>(d, occ, iters, "16 x v_fma_f32");
  run<K_SUB, 8, NACC>(d, occ, iters, "8 x v_sub_f32");
  run<K_AND, 8, NACC>(d, occ, iters, "8 x v_and_b32");
  run<K_PERM, 8, NACC>(d, occ, iters, "8 x v_perm_b32");
  run<K_EXP, 2, NACC>(d, occ, iters, "2 x v_exp_f32");
  run<K_EXP, 4, NACC>(d, occ, iters, "4 x v_exp_f32");
}
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 3000;
  float* d; hipMalloc(&d, 1024 * 256 * 4);
  for (int occ : {1, 3}) { sweep<1>(d, occ, iters); sweep<2>(d, occ, iters); }
  return 0;
}
